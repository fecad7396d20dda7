#!/usr/bin/env bash
# Instruction mix of the frame kernels (developer tool): SQ instruction-class counters of
# tools/one_variant.py frames, in separate --pmc passes (never mixed with trace domains).
#   bash tools/prof_mix.sh <tag> <config> <W> <H> [lib.so]     -> gpurun_out/mix_<tag>/
set -e
TAG=$1; CFG=${2:-c4}; W=${3:-3840}; H=${4:-2160}; LIB=$5
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mix_$TAG
mkdir -p $OUT
if [ -n "$LIB" ]; then export ESC_LIB_PATH=$R/$LIB; fi
export TMPDIR=/tmp
cd /tmp
pass() { # name counters...
  local n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$n -o pmc -- \
    python3 $R/tools/one_variant.py 2 auto 1 $CFG $W $H > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; return 1; }
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass mix1 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
pass mix2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64
pass busy SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
python3 $R/tools/summarize_prof.py $OUT $OUT/summary > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
