"""Developer tool: upload, first-frame, steady-frame and moving-camera times of c3 / c4 / c5."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import torch

import esctp1raytracer_amd as esc
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
    e[0].record(st); r.render_rows(cam, W, H, 0, H, out_f32=buf, flags=esc.ESC_RENDER_NO_COUNTERS); e[1].record(st)   # builds every list
    r.render_rows(cam, W, H, 0, H, out_f32=buf, flags=esc.ESC_RENDER_NO_COUNTERS); e[2].record(st); r.synchronize()
    first, steady = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    mv = []
    for i in range(10):  # a camera that moves every frame: per-camera tables + tile lists rebuilt
        c2 = esc.Camera.for_image((eye[0] + 0.01 * (i + 1), eye[1], eye[2]), look, W, H)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); r.render_rows(c2, W, H, 0, H, out_f32=buf, flags=esc.ESC_RENDER_NO_COUNTERS); b.record(st); r.synchronize()
        mv.append(a.elapsed_time(b))
    mv.sort()
    print(f"{cfg}: upload (stage + commit incl. group build) {1e3 * (t1 - t0):.1f} ms; first frame (per-camera tables, "
          f"tile lists, light lists) {first:.3f} ms; steady frame {steady:.3f} ms; moving camera (tables + tile lists "
          f"rebuilt every frame) {mv[len(mv) // 2]:.3f} ms")
