import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import esctp1raytracer_amd as esc
r = esc.Renderer(0)
sc = esc.Scene.synthetic("c5")
r.upload(sc)
W, H = 640, 360
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
r.render_rows(cam, W, H, 0, H, out_f32=buf); r.synchronize()
L = r.tile_lists(3)
R = L["tiles_x"]
faces = L["counts"].reshape(-1, R, R)
f0 = faces[0]
print("face +x: per-column (u) mean:", [round(float(f0[:, i].mean()), 1) for i in range(0, R, 8)])
print("face +x: per-row (w) mean:", [round(float(f0[i, :].mean()), 1) for i in range(0, R, 8)])
ids = (C.c_int32 * 64)()
cell = (0 * R + 64) * R + 64
n = r._lib.esc_tile_list_ids(r._h, 3, cell, ids, 64)
print("cell (+x, 64, 64): count", n, "ids", list(ids[:min(n, 64)]))
g = sc.geometry(0)
