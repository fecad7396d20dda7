#!/usr/bin/env bash
# Profiles bench.py's default run (c4, 1 GPU) on the GPU box, for BOTH sweeps: culled (the default
# path) and linear (--path linear = ESC_RENDER_INDEX_ORDER).  Separate passes: kernel trace + stats
# of the default command (CPU baseline skipped: it launches no kernels), then PMC counters on their
# own (never mixed with trace domains), without the BVH leg.
# Usage (from the repo root on the GPU box):  bash tools/profile.sh <tag> [config]
set -e
TAG=${1:-r03}
CFG=${2:-c4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for P in culled linear; do
  OUT=$R/gpurun_out/prof_$TAG/$P
  mkdir -p $OUT
  ACC=""; [ $P = linear ] && ACC="--no-accel"
  echo "== [$P] kernel trace + stats"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- \
    python3 $R/bench.py --config $CFG --path $P --steps 10 --warmup 2 --cpu-rows 0 --profile-run $ACC > $OUT/bench_kt.json 2> $OUT/kt.log || { tail -20 $OUT/kt.log; exit 1; }
  pass() { # name counters...
    local n=$1; shift
    echo "== [$P] pmc $n"
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$n -o pmc -- \
      python3 $R/bench.py --config $CFG --path $P --steps 3 --warmup 1 --cpu-rows 0 --no-accel --profile-run > $OUT/bench_$n.json 2> $OUT/$n.log || { tail -20 $OUT/$n.log; return 1; }
  }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
  pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE || echo "(sq2 pass failed, continuing)"
done
find $R/gpurun_out/prof_$TAG -name "*.csv" | head -40
