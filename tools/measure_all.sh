# The end-of-round measurement set in one gpurun call -> gpurun_out/final/ (copied to profiles/rNN_final)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
for c in c2 c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || echo "bench $c failed"
done
for c in "c2 1920 1080" "c3 3840 2160" "c4 3840 2160" "c5 7680 4320"; do
  timeout -k 10 120 python tools/frame_time.py $c 2>&1 | grep -v amdgpu >> $O/frame_times.txt || true
  timeout -k 10 120 python tools/noshadow_time.py $c 2>&1 | grep -v amdgpu >> $O/noshadow_split.txt || true
  timeout -k 10 120 python tools/list_stats.py $c 2>&1 | grep -v amdgpu >> $O/list_stats.txt || true
done
# ESC_STAGE_LDS: the reference-arithmetic A/B path, timed the way the other numbers are
for c in c3 c4; do
  timeout -k 10 300 python bench.py --config $c --stage lds --steps 3 --warmup 1 --cpu-rows 0 --no-accel --linear-steps 0 2> /dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c ESC_STAGE_LDS: %.3f ms per frame, %.1f Mrays/s' % (d['ms_per_step'], d['value']))" >> $O/stage_lds.txt || echo "$c lds failed" >> $O/stage_lds.txt
done
python tools/build_times.py > $O/build_times.txt 2>&1
cat $O/frame_times.txt $O/stage_lds.txt $O/build_times.txt
