"""Render a few frames of one kernel variant (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import esctp1raytracer_amd as esc
px = int(sys.argv[1]) if len(sys.argv) > 1 else 1
stage = {"auto": esc.ESC_STAGE_AUTO, "smem": esc.ESC_STAGE_SMEM, "lds": esc.ESC_STAGE_LDS,
         "bvh": esc.ESC_STAGE_BVH}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
shadows = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
cfg = sys.argv[4] if len(sys.argv) > 4 else "c4"
W, H = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (3840, 2160)
r = esc.Renderer(0)
r.upload(esc.Scene.synthetic(cfg))
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
r.reset_counters()
r.render_rows(cam, W, H, 0, H, out_f32=buf, stage=stage, shadows=shadows, px=px)  # one counted frame
r.synchronize()
c = r.counters()
for _ in range(3):  # ... and three as bench.py times them (ESC_RENDER_NO_COUNTERS)
    r.render_rows(cam, W, H, 0, H, out_f32=buf, stage=stage, shadows=shadows, px=px,
                  flags=esc.ESC_RENDER_NO_COUNTERS)
r.synchronize()
print(c)
if c['anyhit_lane_tests']:
    print('shadow lane efficiency', c['anyhit_tests'] / c['anyhit_lane_tests'])
