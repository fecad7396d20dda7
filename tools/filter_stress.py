"""Hunt for differences between the filtered kernels and the reference arithmetic run for every
pair (ESC_RENDER_EXACT_ONLY), on random scenes, both on the GPU, in both shading forms:
thin and coplanar triangle families, spheres over four decades of radius, cameras inside and
outside, 1-3 lights (two-face lights with the hashed face choice), scenes far from the origin.

python tools/filter_stress.py [n_seeds] [W] [H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import esctp1raytracer_amd as esc

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 160
r = esc.Renderer(0)
bad = 0
lit = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(90000 + seed)
    sc = esc.Scene()
    n_tri = int(rng.integers(1, 3000))
    n_sph = int(rng.integers(0, 3000))
    spread = float(10.0 ** rng.uniform(-0.5, 2.0))
    off = rng.uniform(-1, 1, 3) * (1000.0 if seed % 5 == 4 else 0.0)
    c = rng.uniform(-spread, spread, (n_tri, 1, 3))
    tri = c + rng.normal(0, spread * 10.0 ** rng.uniform(-2.5, -0.3), (n_tri, 3, 3))
    if seed % 3 == 0:  # coplanar, axis-aligned families: grazing rays are likelier
        tri[:, :, 1] = np.round(tri[:, :, 1] / spread * 4) * spread / 4
    if seed % 3 == 1:  # slivers
        tri[::2, 2] = tri[::2, 0] + (tri[::2, 1] - tri[::2, 0]) * rng.uniform(0.2, 0.8, (len(tri[::2]), 1)) \
            + rng.normal(0, spread * 1e-4, (len(tri[::2]), 3))
    tri = (tri + off).astype(np.float32)
    col = np.array([.5, .5, .5] * 2 + [0] * 6 + [10.], np.float32)
    sc.add_geometry(tri.reshape(-1, 3), np.arange(3 * n_tri).reshape(-1, 3), col)
    n_lights = int(rng.integers(1, 4))
    two_faces = seed % 4 == 1
    for _ in range(n_lights):
        p0 = rng.uniform(-spread, spread, 3) + np.array([0, spread, 0]) + off
        lt = np.stack([p0, p0 + rng.normal(0, 0.03 * spread, 3), p0 + rng.normal(0, 0.03 * spread, 3)]).astype(np.float32)
        m = col.copy()
        m[9:12] = (9, 8, 7)
        if two_faces:
            lt = np.concatenate([lt, lt + rng.normal(0, 0.05 * spread, (1, 3)).astype(np.float32)])
            sc.add_geometry(lt, np.array([[0, 1, 2], [3, 4, 5]]), m)
        else:
            sc.add_geometry(lt, np.array([[0, 1, 2]]), m)
    if n_sph:
        sph = np.concatenate([rng.uniform(-spread, spread, (n_sph, 3)) + off,
                              spread * 10.0 ** rng.uniform(-4, -0.5, (n_sph, 1))], 1).astype(np.float32)
        sc.add_spheres(sph, np.tile(col, (n_sph, 1)))
    r.upload(sc)
    eye = rng.uniform(-1.5 * spread, 1.5 * spread, 3) * np.array([1.0, 0.4, 1.0]) + off
    eye[1] += 0.6 * spread  # above the cloud, looking in, so that most frames are lit
    if seed % 7 == 3 and n_sph:  # camera inside a sphere
        eye = sph[0, :3].astype(np.float64) + 0.3 * sph[0, 3]
    look = rng.uniform(-0.3 * spread, 0.3 * spread, 3) + off
    cam = esc.Camera.for_image(tuple(eye), tuple(look), W, H)
    kw = {"face_mode": esc.ESC_FACE_HASH, "seed": seed} if two_faces else {}
    ref = r.render(cam, W, H, flags=esc.ESC_RENDER_EXACT_ONLY | esc.ESC_RENDER_SHADE_FUSED, **kw)
    lit += int(ref.any())
    for name, flags in (("fused", esc.ESC_RENDER_SHADE_FUSED), ("queue", esc.ESC_RENDER_SHADE_QUEUE),
                        ("queue+index", esc.ESC_RENDER_SHADE_QUEUE | esc.ESC_RENDER_INDEX_ORDER)):
        got = r.render(cam, W, H, flags=flags, **kw)
        nd = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
        if nd:
            bad += 1
            print(f"seed {seed} [{name}]: {nd} values differ (n_tri {n_tri}, n_sph {n_sph}, "
                  f"spread {spread:.3g}, lights {n_lights})", flush=True)
    if seed % 100 == 99:
        print(f"... {seed + 1} scenes, {bad} differences so far", flush=True)
print(f"{n_seeds} scenes ({lit} with a lit pixel), {bad} comparisons with differences")
