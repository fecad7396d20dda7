# kernel trace + two PMC passes of the ESC_STAGE_BVH frame (c4 by default)
set -e
R=$GRAFT_REPO_ROOT
CFG=${1:-c4}; W=${2:-3840}; H=${3:-2160}; TAG=${4:-bvh}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/one_variant.py 1 bvh 1 $CFG $W $H > $O/kt.log 2>&1 || tail -5 $O/kt.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o pmc -- python3 $R/tools/one_variant.py 1 bvh 1 $CFG $W $H > $O/sq.log 2>&1 || tail -5 $O/sq.log
# FETCH_SIZE and WRITE_SIZE need separate passes (together rocprofv3 aborts and the run hangs)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 $R/tools/one_variant.py 1 bvh 1 $CFG $W $H > $O/fetch.log 2>&1 || tail -5 $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 $R/tools/one_variant.py 1 bvh 1 $CFG $W $H > $O/write.log 2>&1 || tail -5 $O/write.log
python3 - <<PY
import csv, collections, glob
print(open(glob.glob("$O/kt/**/kt_kernel_stats.csv", recursive=True)[0]).read())
for d in ("pmc_sq","pmc_fetch","pmc_write"):
    f=glob.glob("$O/%s/**/pmc_counter_collection.csv"%d, recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(d, k, {c: "%.4g"%(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
