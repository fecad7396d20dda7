"""Developer timing loop: brute-force stage x pixels-per-lane variants on the BASELINE c4 workload (one process, interleaved
rounds, HIP events on the stream the kernels run on)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import esctp1raytracer_amd as esc

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
n = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
st = torch.cuda.Stream()
r = esc.Renderer(0, stream=st)
sc = esc.Scene.synthetic(cfg, n)
print(sc.info())
r.upload(sc)
eye, look = esc.synthetic_view()
cam = esc.Camera.for_image(eye, look, W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
variants = []
for px in (1, 2, 4):  # pixels per work-item of the primary pass
    variants += [(f"smem px{px}", esc.ESC_STAGE_SMEM, True, px), (f"lds px{px}", esc.ESC_STAGE_LDS, True, px),
                 (f"smem px{px} noshadow", esc.ESC_STAGE_SMEM, False, px),
                 (f"lds px{px} noshadow", esc.ESC_STAGE_LDS, False, px)]
res = {v[0]: [] for v in variants}
for rd in range(rounds + 1):
    for name, stage, sh, px in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r.reset_counters()
        with torch.cuda.stream(st):
            e0.record(st)
            r.render_rows(cam, W, H, 0, H, out_f32=buf, stage=stage, shadows=sh, px=px)
            e1.record(st)
        st.synchronize()
        if rd:
            res[name].append(e0.elapsed_time(e1))
c = r.counters()
rays = c["primary_rays"] + c["shadow_rays"]
print("counters(last variant)", c)
for k, v in res.items():
    v.sort()
    print(f"{k:22s} min {v[0]:9.3f} ms  med {v[len(v)//2]:9.3f} ms")
