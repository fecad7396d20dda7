"""Developer tool: the host-buffer frame (esc_render_frame_host: render + copy back, synchronous), into a
fresh numpy array and into a reused one.  python tools/host_frame_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import esctp1raytracer_amd as esc

for cfg, W, H in (("c4", 3840, 2160), ("c5", 7680, 4320)):
    r = esc.Renderer(0)
    r.upload(esc.Scene.synthetic(cfg))
    cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
    buf = np.zeros((H, W, 3), np.float32)
    buf[:] = 1.0  # touched
    for name, kw in (("fresh array per frame", {}), ("reused array", {"out": buf})):
        ts = []
        for i in range(6):
            t0 = time.perf_counter()
            r.render(cam, W, H, flags=esc.ESC_RENDER_NO_COUNTERS, **kw)
            ts.append(time.perf_counter() - t0)
        print(f"{cfg} esc_render_frame_host, {name}: {1e3 * sorted(ts)[len(ts) // 2]:.2f} ms per frame "
              f"({W * H * 12 / 1e6:.0f} MB copied back)")
    r.close()
