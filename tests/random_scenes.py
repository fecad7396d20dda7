"""Seeded random scenes for the list hunts (tests/test_gpu_lists.py::test_random_scenes_*, tools/list_hunt.py):
sphere clouds of every density, heightfield patches and loose triangles, one to three one-face lights
anywhere (inside the geometry included), cameras anywhere (inside included), odd image sizes."""
import numpy as np

import oracle_lib as ol


def _patch(rng, n, size, centre, rough):
    xs = np.linspace(-size, size, n + 1) + centre[0]
    zs = np.linspace(-size, size, n + 1) + centre[2]
    hgt = centre[1] + rough * rng.standard_normal((n + 1, n + 1))
    tris = []
    for j in range(n):
        for i in range(n):
            a = (xs[i], hgt[j, i], zs[j])
            b = (xs[i + 1], hgt[j, i + 1], zs[j])
            c = (xs[i + 1], hgt[j + 1, i + 1], zs[j + 1])
            e = (xs[i], hgt[j + 1, i], zs[j + 1])
            tris += [a, b, c, a, c, e]
    return np.array(tris, np.float32)


def random_scene(seed):
    """-> (scene dict, eye, look, W, H, vfov)"""
    rng = np.random.default_rng(seed)
    box = 10.0 ** rng.uniform(0.0, 1.5)  # scenes of 1 .. 30 units
    origin = rng.uniform(-1, 1, 3) * (0.0 if seed % 3 else 10.0 ** rng.uniform(0, 3))  # sometimes far from 0
    geoms = []
    kind = seed % 4
    n_sph = 0
    if kind in (0, 1, 3):
        n_sph = int(10.0 ** rng.uniform(1.0, 3.3))
    flat_y = None
    if kind in (1, 2, 3):
        n = int(rng.integers(3, 40 if kind == 2 else 14))
        c = origin + rng.uniform(-0.3, 0.3, 3) * box
        rough = box * 10.0 ** rng.uniform(-3, -0.7)
        if seed % 9 == 0:  # a flat patch, the camera (below) nearly in its plane: escape entries
            rough = 0.0 if seed % 18 == 0 else box * 10.0 ** rng.uniform(-7, -4)
            flat_y = float(c[1])
        geoms.append({"vertex": _patch(rng, n, box * rng.uniform(0.3, 1.0), c, rough),
                      "face_index": np.arange(6 * n * n).reshape(-1, 3), "material": ol.WHITE})
    if kind == 3 or seed % 5 == 0:  # loose triangles of every size, slivers included
        k = int(rng.integers(4, 200))
        p = origin + rng.uniform(-1, 1, (k, 1, 3)) * box
        e = rng.standard_normal((k, 3, 3)) * box * 10.0 ** rng.uniform(-2.5, -0.3, (k, 1, 1))
        if seed % 10 == 0:
            e[:, 2] = e[:, 1] * rng.uniform(0.99, 1.01, (k, 1)) + e[:, 0] * 1e-4  # slivers
        geoms.append({"vertex": (p + e).reshape(-1, 3).astype(np.float32),
                      "face_index": np.arange(3 * k).reshape(-1, 3), "material": ol.RED})
    n_lights = 1 + (seed % 7 == 0) + (seed % 11 == 0)
    for li in range(n_lights):
        where = seed % 6
        lp = origin + (rng.uniform(-0.5, 0.5, 3) * box if where == 0 else          # inside the geometry
                       np.array([rng.uniform(-1, 1), rng.uniform(1.0, 3.0), rng.uniform(-1, 1)]) * box)
        s = box * 10.0 ** rng.uniform(-2, -0.5)
        tri = lp + np.array([(-s, 0, 0), (s, 0, 0), (0, 0, -s)])
        geoms.append({"vertex": tri.astype(np.float32), "face_index": np.array([[0, 1, 2]]),
                      "material": ol.LIGHT_A})
    sph = mats = None
    if n_sph:
        c = origin + rng.uniform(-1, 1, (n_sph, 3)) * box * rng.uniform(0.2, 1.0)
        rad = box * 10.0 ** rng.uniform(-2.5, -0.6, n_sph) * rng.uniform(0.3, 1.0)
        if seed % 8 == 0:
            rad[: max(1, n_sph // 50)] *= 8.0  # a few large ones (cameras and lights end up inside them)
        sph = np.concatenate([c, rad[:, None]], axis=1).astype(np.float32)
        cols = rng.uniform(0.2, 0.9, (n_sph, 3))
        mats = np.stack([ol.material13(ka=k, kd=k) for k in cols])
    d = ol.scene_dict(geoms, sph, mats)
    eye = origin + rng.uniform(-1.2, 1.2, 3) * box * (0.6 if seed % 2 else 1.5)
    look = origin + rng.uniform(-0.5, 0.5, 3) * box
    if flat_y is not None:  # within 1e-7 .. 1e-3 of the scene's size of the patch's plane, looking along it
        off = box * 10.0 ** rng.uniform(-7, -3) * rng.choice([-1.0, 1.0])
        eye[1] = flat_y + off
        look[1] = flat_y + off * rng.uniform(-2, 2)
    if np.linalg.norm(look - eye) < 1e-3 * box:
        look = eye + np.array([0.0, 0.0, -box])
    W = int(rng.integers(33, 300))
    H = int(rng.integers(9, 200))
    vfov = float(rng.choice([5.0, 30.0, 60.0, 100.0, 150.0]))
    return d, tuple(float(x) for x in eye), tuple(float(x) for x in look), W, H, vfov
