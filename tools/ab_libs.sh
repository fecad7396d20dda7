#!/usr/bin/env bash
# Developer A/B: per-kernel splits for configs and libraries: bash tools/ab_libs.sh <out> "<cfgs>" lib...
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; CFGS=$2; shift; shift
mkdir -p $OUT
cd $R
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = default ]; then unset ESC_LIB_PATH; else export ESC_LIB_PATH=$R/$lib; fi
  for c in $CFGS; do
    case $c in c5) a="c5 7680 4320";; *) a="$c 3840 2160";; esac
    echo -n "[$lib] " >> $OUT/splits.log
    timeout -k 10 120 python tools/split_time.py $a 2>&1 | grep -v amdgpu.ids >> $OUT/splits.log
  done
done
done
cat $OUT/splits.log
