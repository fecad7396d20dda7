#!/usr/bin/env bash
# Profiles bench.py's default run (c4, 1 GPU) on the GPU box.  Separate passes: kernel trace +
# stats of the default command (CPU baseline skipped: it launches no kernels), then PMC counters
# on their own (never mixed with trace domains), without the BVH leg.
# Usage (from the repo root on the GPU box):  bash tools/profile.sh <tag> [config]
set -e
TAG=${1:-r02}
CFG=${2:-c4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- \
  python3 $R/bench.py --config $CFG --steps 10 --warmup 2 --cpu-rows 0 --profile-run > $OUT/bench_kt.json 2> $OUT/kt.log || { tail -20 $OUT/kt.log; exit 1; }
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- \
  python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --cpu-rows 0 --no-accel --profile-run > $OUT/bench_fetch.json 2> $OUT/fetch.log || { tail -20 $OUT/fetch.log; exit 1; }
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- \
  python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --cpu-rows 0 --no-accel --profile-run > $OUT/bench_write.json 2> $OUT/write.log || { tail -20 $OUT/write.log; exit 1; }
echo "== pmc SQ"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- \
  python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --cpu-rows 0 --no-accel --profile-run > $OUT/bench_sq.json 2> $OUT/sq.log || { tail -20 $OUT/sq.log; exit 1; }
echo "== pmc SQ2"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- \
  python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --cpu-rows 0 --no-accel --profile-run > $OUT/bench_sq2.json 2> $OUT/sq2.log || { tail -20 $OUT/sq2.log; echo "(sq2 pass failed, continuing)"; }
find $OUT -name "*.csv" | head -30
