"""Developer tool: render_rows vs render_strips(0, 1) vs a recorded frame, same scene and camera."""
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
bufs = [torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0") for _ in range(3)]
rec = [None]


def launch_recorded(i):
    if i == 0:
        rec[0] = r.record_strips(cam, W, H, 0, 1, out_f32=bufs[0])
    rec[0].launch()


def timed(fn, n=20):
    ms = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn(i)
        e1.record(st)
        st.synchronize()
        if i >= 3:
            ms.append(e0.elapsed_time(e1))
    ms.sort()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(n):
        fn(i)
    e1.record(st)
    st.synchronize()
    return ms[len(ms) // 2], e0.elapsed_time(e1) / n


for rd in range(2):
    for name, fn in (("render_rows", lambda i: r.render_rows(cam, W, H, 0, H, out_f32=bufs[0])),
                     ("render_strips(0,1)", lambda i: r.render_strips(cam, W, H, 0, 1, out_f32=bufs[0])),
                     ("render_strips, 3 buffers", lambda i: r.render_strips(cam, W, H, 0, 1, out_f32=bufs[i % 3])),
                     ("recorded frame", launch_recorded)):
        a, b = timed(fn)
        print(f"{cfg} {name:26s}: frame {a:.3f} ms, back to back {b:.3f} ms")
