"""Developer tool: what the tile lists of the primary pass look like on a config."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import esctp1raytracer_amd as esc

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
r = esc.Renderer(0)
r.upload(esc.Scene.synthetic(cfg))
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
r.render_rows(cam, W, H, 0, H, out_f32=buf)
r.synchronize()
for which, name in ((0, "spheres"), (1, "triangles"), (2, "light-list cells (sphere pairs)"), (3, "light-list cells (triangle pairs)")):
    L = r.tile_lists(which)
    if L is None:
        print(cfg, name, ": no lists")
        continue
    c = L["counts"]
    print(cfg, name, {k: v for k, v in L.items() if k != "counts"})
    print("  tiles", c.size, "empty", int((c == 0).sum()), "overflowing", int((c > L["cap"]).sum()),
          "mean", float(c.mean()), "mean of non-empty", float(c[c > 0].mean()) if (c > 0).any() else 0,
          "max", int(c.max()))
    print("  histogram (0,1-4,5-8,9-16,17-32,33-64,65+):",
          [int(((c >= a) & (c <= b)).sum()) for a, b in ((0, 0), (1, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 1 << 30))])
    rows = c.mean(axis=1)
    print("  mean per tile row, 16 bands top->bottom of the band:", [round(float(x), 1) for x in
                                                                     [rows[i * len(rows) // 16:(i + 1) * len(rows) // 16].mean() for i in range(16)]])
    if which >= 2:
        R = L["tiles_x"]
        faces = c.reshape(-1, R, R)
        print("  per face (+x -x +y -y +z -z per light): mean", [round(float(f.mean()), 2) for f in faces],
              "max", [int(f.max()) for f in faces])
