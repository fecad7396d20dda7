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
python - <<'PY' > $O/build_times.txt 2>&1
import time, torch, esctp1raytracer_amd as esc
st = torch.cuda.Stream()
r = esc.Renderer(0, stream=st)   # events below are recorded on the stream the kernels run on
for cfg, W, H in (("c3", 3840, 2160), ("c4", 3840, 2160), ("c5", 7680, 4320)):
    sc = esc.Scene.synthetic(cfg)
    r.upload(sc); r.synchronize()
    t0 = time.perf_counter(); r.upload(sc); r.synchronize(); t1 = time.perf_counter()
    eye, look = esc.synthetic_view()
    cam = esc.Camera.for_image(eye, look, W, H)
    buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(st); r.render_rows(cam, W, H, 0, H, out_f32=buf); e[1].record(st)   # builds every list
    r.render_rows(cam, W, H, 0, H, out_f32=buf); e[2].record(st); r.synchronize()
    first, steady = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    mv = []
    for i in range(10):  # a camera that moves every frame: per-camera tables + tile lists rebuilt
        c2 = esc.Camera.for_image((eye[0] + 0.01 * (i + 1), eye[1], eye[2]), look, W, H)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); r.render_rows(c2, W, H, 0, H, out_f32=buf); b.record(st); r.synchronize()
        mv.append(a.elapsed_time(b))
    mv.sort()
    print(f"{cfg}: upload (stage + commit incl. group build) {1e3 * (t1 - t0):.1f} ms; first frame (per-camera tables, "
          f"tile lists, light lists) {first:.3f} ms; steady frame {steady:.3f} ms; moving camera (tables + tile lists "
          f"rebuilt every frame) {mv[len(mv) // 2]:.3f} ms")
PY
cat $O/frame_times.txt $O/stage_lds.txt $O/build_times.txt
