"""Developer tool: whole-frame times (HIP events on the render stream, one frame at a time and 20
frames back to back) of a config for several flag sets:  python tools/frame_time.py c4 3840 2160"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import esctp1raytracer_amd as esc

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
st = torch.cuda.Stream()
r = esc.Renderer(0, stream=st)
r.upload(esc.Scene.synthetic(cfg))
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
NC = esc.ESC_RENDER_NO_COUNTERS  # what bench.py times: the ray counters are instrumentation
variants = [("one kernel (default)", 0), ("two kernels", esc.ESC_RENDER_TWO_KERNELS),
            ("one kernel, no lists", esc.ESC_RENDER_NO_TILE_LISTS | esc.ESC_RENDER_NO_LIGHT_LISTS),
            ("two kernels, no lists", esc.ESC_RENDER_TWO_KERNELS | esc.ESC_RENDER_NO_TILE_LISTS |
             esc.ESC_RENDER_NO_LIGHT_LISTS)]
sums = {}
for rd in range(3):
    for name, flags in variants:
        ms = []
        with torch.cuda.stream(st):
            for i in range(12):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                r.render_rows(cam, W, H, 0, H, out_f32=buf, flags=flags | NC)
                e1.record(st)
                st.synchronize()
                if i >= 2:
                    ms.append(e0.elapsed_time(e1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(20):
                r.render_rows(cam, W, H, 0, H, out_f32=buf, flags=flags | NC)
            e1.record(st)
            st.synchronize()
        ms.sort()
        sums.setdefault(name, []).append((ms[len(ms) // 2], e0.elapsed_time(e1) / 20, float(buf.double().sum().item())))
for name, v in sums.items():
    v.sort()
    print(f"{cfg} {name:26s}: frame {v[1][0]:.3f} ms, back to back {v[1][1]:.3f} ms, checksum {v[1][2]:.6f}")
