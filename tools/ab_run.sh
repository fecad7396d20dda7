#!/usr/bin/env bash
# Developer A/B: per-kernel splits of c3/c4/c5 for each library given (paths relative to the repo
# root; "default" = the in-tree build).  Usage on the GPU box: bash tools/ab_run.sh <outdir> lib...
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = default ]; then unset ESC_LIB_PATH; else export ESC_LIB_PATH=$R/$lib; fi
  for c in "c3 3840 2160" "c4 3840 2160" "c5 7680 4320"; do
    echo -n "[$lib] " >> $OUT/splits.log
    timeout -k 10 120 python tools/split_time.py $c >> $OUT/splits.log 2>&1
  done
done
done
cat $OUT/splits.log
