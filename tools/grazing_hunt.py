"""Developer hunt: does the tree's HEURISTIC triangle pad (ESC_RENDER_BVH_HEURISTIC_PADS) ever cull a hit
the reference reports?  Targeted at the case DESIGN.md 4b names: small triangles far from the
camera, the camera nearly in their plane, rays a few 1e-3 rad off the plane, so that the reference's
rounding noise accepts hits whose plane point lies outside the triangle by more than the pad.
Every frame: tree (flag) vs the default path (proven), bit for bit.  python tools/grazing_hunt.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import esctp1raytracer_amd as esc
import oracle_lib as ol

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r = esc.Renderer(0)
W, H = 256, 96
found = 0
with_hits = 0
grazing_hits = 0
for it in range(n):
    dist = 10.0 ** rng.uniform(1.0, 2.7)          # 10 .. 500 units away
    size = 10.0 ** rng.uniform(-1.0, 0.7)         # triangles of 0.1 .. 5 units
    k = int(rng.integers(6, 40))
    tris = []
    for _ in range(k):  # nearly coplanar small triangles scattered around (0, 0, -dist), plane y = 0
        c = np.array([rng.uniform(-0.2, 0.2) * dist, 0.0, -dist * rng.uniform(0.7, 1.3)])
        a = c + np.array([rng.uniform(-1, 1), rng.uniform(-1e-3, 1e-3), rng.uniform(-1, 1)]) * size
        b = c + np.array([rng.uniform(-1, 1), rng.uniform(-1e-3, 1e-3), rng.uniform(-1, 1)]) * size
        tris += [c, a, b]
    geoms = [{"vertex": np.array(tris, np.float32), "face_index": np.arange(3 * k).reshape(-1, 3),
              "material": ol.WHITE},
             {"vertex": np.array([(-1, 0.3 * dist, -dist), (1, 0.3 * dist, -dist), (0, 0.3 * dist, -dist - 1)], np.float32),
              "face_index": np.array([[0, 1, 2]]), "material": ol.LIGHT_A}]
    d = ol.scene_dict(geoms)
    eye = (float(rng.uniform(-1, 1)), float(dist * 10.0 ** rng.uniform(-4.5, -2.0) * rng.choice([-1, 1])), 0.0)
    look = (0.0, 0.0, -dist)
    r.upload(ol.scene_to_product(d))
    cam = esc.Camera.for_image(eye, look, W, H, vfov=float(rng.uniform(5, 40)))
    a = r.render(cam, W, H, shadows=bool(it % 2))
    b = r.render(cam, W, H, shadows=bool(it % 2), stage=esc.ESC_STAGE_BVH, flags=esc.ESC_RENDER_BVH_HEURISTIC_PADS)
    nd = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
    c = r.counters()
    with_hits += 1 if a.any() else 0
    if nd:
        found += 1
        print(f"DIFFERENCE it={it} dist={dist:.3g} size={size:.3g} eye={eye} pixels={nd} hit pixels={int((a.sum(axis=2) > 0).sum())}")
        if found <= 3:
            np.savez(os.path.join(ROOT, "gpurun_out", f"grazing_case_{it}.npz"), tris=np.array(tris, np.float32),
                     eye=np.array(eye), look=np.array(look), dist=dist, W=W, H=H)
print(f"{n} scenes ({with_hits} with lit pixels), {found} with a tree-vs-proven difference")
