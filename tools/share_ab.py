"""Developer tool: GPU time of rank r's share of an N-rank frame for several strip heights (HIP events,
frames back to back):  python tools/share_ab.py c4 8"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import esctp1raytracer_amd as esc
from esctp1raytracer_amd import multigpu

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = {"c2": (1920, 1080), "c3": (3840, 2160), "c4": (3840, 2160), "c5": (7680, 4320)}[cfg]
st = torch.cuda.Stream()
r = esc.Renderer(0, stream=st)
r.upload(esc.Scene.synthetic(cfg))
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda:0")


def timed(fn, n=40):
    for _ in range(3):
        fn()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


full = timed(lambda: r.render_strips(cam, W, H, 0, 1, out_u8=buf, shadows=cfg != "c2", flags=esc.ESC_RENDER_NO_COUNTERS))
print(f"{cfg} whole frame {full:.1f} us; ideal share of {N}: {full / N:.1f} us")
for name, fl in (("one kernel", 0), ("two kernels", esc.ESC_RENDER_TWO_KERNELS)):
    for S in (8, 32):
        ts = [timed(lambda: r.render_strips(cam, W, H, k, N, out_u8=buf, shadows=cfg != "c2", strip_rows=S, flags=fl | esc.ESC_RENDER_NO_COUNTERS))
              for k in range(N)]
        print(f"  {name}, strips of {S:2d} rows: rank shares " + " ".join(f"{t:.1f}" for t in ts) +
              f"  max {max(ts):.1f} us")
