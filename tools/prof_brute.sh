# kernel trace + one SQ PMC pass of the brute-force frame of a config (default c5 at 8K)
set -e
R=$GRAFT_REPO_ROOT
CFG=${1:-c5}; W=${2:-7680}; H=${3:-4320}; TAG=${4:-brute_c5}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/one_variant.py 2 auto 1 $CFG $W $H > $O/kt.log 2>&1 || tail -5 $O/kt.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o pmc -- python3 $R/tools/one_variant.py 2 auto 1 $CFG $W $H > $O/sq.log 2>&1 || tail -5 $O/sq.log
python3 $R/tools/summarize_prof.py $O $O/summary
head -4 $O/summary/kernel_stats.csv
