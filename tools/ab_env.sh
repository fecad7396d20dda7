#!/usr/bin/env bash
# Developer A/B: per-kernel splits of c3/c4/c5 for each environment setting given ("-" = none),
# e.g.  bash tools/ab_env.sh r3b - ESC_LISTS=0
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in "$@"; do
  for c in "c3 3840 2160" "c4 3840 2160" "c5 7680 4320"; do
    if [ "$v" = - ]; then timeout -k 10 120 python tools/split_time.py $c >> $OUT/splits.log 2>&1
    else env $v timeout -k 10 120 python tools/split_time.py $c >> $OUT/splits.log 2>&1; fi
  done
done
done
grep -v amdgpu.ids $OUT/splits.log
