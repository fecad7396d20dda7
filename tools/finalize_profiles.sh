#!/usr/bin/env bash
# After tools/measure_all.sh + tools/profile.sh <tag> c4 (+ tools/prof_brute.sh c5 ... <tag>_c5) ran on the GPU
# box: condense gpurun_out/ into profiles/<tag>_final and stamp profiles/current.json.
#   bash tools/finalize_profiles.sh r03
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
D=profiles/${TAG}_final
mkdir -p $D
python tools/summarize_prof.py gpurun_out/prof_$TAG $D --current c4 > /dev/null
cp gpurun_out/prof_$TAG/culled/bench_kt.json $D/bench_under_rocprof.json 2>/dev/null || true
for f in bench_c2.json bench_c3.json bench_c4.json bench_c5.json frame_times.txt noshadow_split.txt list_stats.txt \
         build_times.txt stage_lds.txt rank_shares.txt strip_latency.txt bench_c4_2ranks_gloo_one_gpu.json; do
  [ -f gpurun_out/final/$f ] && cp gpurun_out/final/$f $D/$f
done
if [ -d gpurun_out/prof_${TAG}_c5/summary ]; then
  mkdir -p $D/c5 && cp gpurun_out/prof_${TAG}_c5/summary/* $D/c5/
fi
ls $D
