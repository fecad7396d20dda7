"""Developer timing: brute force (AUTO) vs ESC_STAGE_BVH on the BASELINE configs, one process,
HIP events on the render stream; also checks the two frames are identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import esctp1raytracer_amd as esc

cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c3", "c4", "c5"]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
brute = (len(sys.argv) <= 3) or sys.argv[3] != "nobrute"
bvh = (len(sys.argv) <= 3) or sys.argv[3] != "nobvh"
SIZE = {"c2": (1920, 1080), "c3": (3840, 2160), "c4": (3840, 2160), "c5": (7680, 4320)}
st = torch.cuda.Stream()
r = esc.Renderer(0, stream=st)
eye, look = esc.synthetic_view()
for cfg in cfgs:
    W, H = SIZE[cfg]
    sc = esc.Scene.synthetic(cfg)
    r.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    shadows = cfg != "c2"
    bufs = {}
    for name, stage in (("brute", esc.ESC_STAGE_AUTO), ("bvh", esc.ESC_STAGE_BVH)):
        if (name == "brute" and not brute) or (name == "bvh" and not bvh):
            continue
        buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
        ts = []
        for rd in range(rounds + 1):
            if name == "brute" and cfg == "c5" and rd > 1:
                break
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r.reset_counters()
            with torch.cuda.stream(st):
                e0.record(st)
                r.render_rows(cam, W, H, 0, H, out_f32=buf, stage=stage, shadows=shadows)
                e1.record(st)
            st.synchronize()
            if rd:
                ts.append(e0.elapsed_time(e1))
        c = r.counters()
        rays = c["primary_rays"] + c["shadow_rays"]
        ts.sort()
        bufs[name] = buf
        print(f"{cfg} {name:6s} min {ts[0]:9.3f} ms  med {ts[len(ts)//2]:9.3f} ms  "
              f"{rays / ts[0] / 1e3:10.1f} Mrays/s  tests/shadow-ray "
              f"{c['anyhit_tests'] / max(c['shadow_rays'], 1):8.1f} lane-eff "
              f"{c['anyhit_tests'] / max(c['anyhit_lane_tests'], 1):.3f}", flush=True)
    if len(bufs) == 2:
        nd = int((bufs["brute"].view(torch.int32) != bufs["bvh"].view(torch.int32)).sum().item())
        print(f"{cfg} differing fp32 values: {nd}")
    if bvh:
        print(cfg, "accel", r.accel_info(), flush=True)
