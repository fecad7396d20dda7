"""Per-kernel split (k_primary / shading, HIP events inside the library) of one workload, default
path, N frames; run once per environment variant:
    ESC_GROUP_SEG=2048 python tools/split_time.py c4 3840 2160 [flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import esctp1raytracer_amd as esc

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
r = esc.Renderer(0)
r.upload(esc.Scene.synthetic(cfg))
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
ms = []
for i in range(13):
    r.reset_counters()
    r.render_rows(cam, W, H, 0, H, out_f32=buf, flags=flags | esc.ESC_RENDER_TIME_KERNELS)
    r.synchronize()
    if i >= 3:
        ms.append(r.last_kernel_ms())
c = r.counters()
a = sorted(m[0] for m in ms)[len(ms) // 2]
b = sorted(m[1] for m in ms)[len(ms) // 2]
eff = c["anyhit_tests"] / c["anyhit_lane_tests"] if c["anyhit_lane_tests"] else 0
env = {k: v for k, v in os.environ.items() if k.startswith("ESC_")}
print(f"{cfg} {W}x{H} flags {flags} {env}: k_primary {a:.3f} ms, shading {b:.3f} ms, frame {a + b:.3f} ms, "
      f"lane efficiency {eff:.3f}, checksum {float(buf.double().sum().item()):.6f}")
