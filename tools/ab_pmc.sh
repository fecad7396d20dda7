set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for v in work head; do
  D=$R; [ $v = head ] && D=$R/build/ab_head
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/ab_$v -o pmc -- python3 $D/tools/one_variant.py 1 smem 1 > $R/gpurun_out/ab_$v.log 2>&1 || tail -5 $R/gpurun_out/ab_$v.log
done
python3 - <<PY
import csv, collections, glob
for v in ("work","head"):
    f=glob.glob("$R/gpurun_out/ab_%s/**/pmc_counter_collection.csv"%v, recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta=(r["VGPR_Count"],r["SGPR_Count"],r["Scratch_Size"])
    print(v, meta, {k: "%.4g"%(sum(x)/len(x)) for k,x in sorted(agg.items())})
PY
