"""One-off hunt for BVH-vs-brute-force differences on random scenes (both on the GPU, no oracle):
python tools/bvh_stress.py [n_seeds] [W] [H]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import esctp1raytracer_amd as esc

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W = int(sys.argv[2]) if len(sys.argv) > 2 else 320
H = int(sys.argv[3]) if len(sys.argv) > 3 else 200
r = esc.Renderer(0)
bad = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(50000 + seed)
    sc = esc.Scene()
    n_tri = int(rng.integers(1, 3000))
    n_sph = int(rng.integers(0, 2000))
    spread = rng.uniform(1, 20)
    c = rng.uniform(-spread, spread, (n_tri, 1, 3))
    tri = (c + rng.normal(0, rng.uniform(0.01, 2.0), (n_tri, 3, 3))).astype(np.float32)
    if seed % 3 == 0:  # axis-aligned / coplanar families: grazing rays are likelier
        tri[:, :, 1] = np.round(tri[:, :, 1])
    col = np.array([.5, .5, .5] * 2 + [0] * 6 + [10.], np.float32)
    sc.add_geometry(tri.reshape(-1, 3), np.arange(3 * n_tri).reshape(-1, 3), col)
    n_lights = int(rng.integers(1, 4))
    two_faces = seed % 4 == 1  # two-face lights: hashed face choice, 2 sample points per light
    for _ in range(n_lights):
        p0 = rng.uniform(-spread, spread, 3) + np.array([0, spread, 0])
        lt = np.stack([p0, p0 + rng.normal(0, 0.3, 3), p0 + rng.normal(0, 0.3, 3)]).astype(np.float32)
        m = col.copy(); m[9:12] = (9, 8, 7)
        if two_faces:
            lt = np.concatenate([lt, lt + rng.normal(0, 0.5, (1, 3)).astype(np.float32)])
            sc.add_geometry(lt, np.array([[0, 1, 2], [3, 4, 5]]), m)
        else:
            sc.add_geometry(lt, np.array([[0, 1, 2]]), m)
    if n_sph:
        sph = np.concatenate([rng.uniform(-spread, spread, (n_sph, 3)),
                              rng.uniform(0.01, 0.1 * spread, (n_sph, 1))], 1).astype(np.float32)
        sc.add_spheres(sph, np.tile(col, (n_sph, 1)))
    r.upload(sc)
    eye = rng.uniform(-1.5 * spread, 1.5 * spread, 3)
    look = rng.uniform(-0.3 * spread, 0.3 * spread, 3)
    cam = esc.Camera.for_image(tuple(eye), tuple(look), W, H)
    kw = {"face_mode": esc.ESC_FACE_HASH, "seed": seed} if two_faces else {}
    a = r.render(cam, W, H, **kw)
    b = r.render(cam, W, H, stage=esc.ESC_STAGE_BVH, **kw)
    nd = int((a.view(np.uint32) != b.view(np.uint32)).sum())
    if nd:
        bad += 1
        print(f"seed {seed}: {nd} values differ (n_tri {n_tri}, n_sph {n_sph}, spread {spread:.2f})", flush=True)
print(f"{n_seeds} scenes, {bad} with differences")
