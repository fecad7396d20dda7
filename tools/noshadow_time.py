"""Developer tool: per-kernel split of a config with and without shadow rays (what the shading
kernel costs apart from its any-hit sweeps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import esctp1raytracer_amd as esc

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
r = esc.Renderer(0)
r.upload(esc.Scene.synthetic(cfg))
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
for sh in (True, False):
    ms = []
    for i in range(13):
        r.render_rows(cam, W, H, 0, H, out_f32=buf, shadows=sh, flags=esc.ESC_RENDER_TIME_KERNELS | esc.ESC_RENDER_NO_COUNTERS)
        r.synchronize()
        if i >= 3:
            ms.append(r.last_kernel_ms())
    a = sorted(m[0] for m in ms)[len(ms) // 2]
    b = sorted(m[1] for m in ms)[len(ms) // 2]
    print(f"{cfg} shadows={sh}: k_primary {a:.3f} ms, shading {b:.3f} ms")
