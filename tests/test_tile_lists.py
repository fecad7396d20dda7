"""CPU checks of the geometry behind the tile lists of the primary pass (csrc/rt_tile_math.h,
reached through esc_tile_rect / esc_tile_cone -- the very code the binning kernels run).

A tile list replaces the question "which leaf groups can a ray of this tile touch"; it is complete
iff (S) every pixel whose ray -- with the direction the reference computes in fp32 (main.cpp:709-713,
camera.h:31-34) -- passes within R of the centre lies inside the rectangle the group is appended to,
and (E) every ray direction of a tile lies within the tile's chord delta of its centre direction.
Both are checked here per pixel by brute force, on cameras near and far from the world origin, skewed
image planes, spheres in front of, beside and behind the camera, from specks to spheres that fill
the screen.
"""
import ctypes as C
import zlib

import numpy as np
import pytest

import esctp1raytracer_amd as esc
from esctp1raytracer_amd import _capi

F32 = np.float32


def ref_dirs(cam, W, H):
    """unit directions of every pixel exactly as the reference computes them (fp32, same order)"""
    o = np.array(cam.origin, F32)
    llc = np.array(cam.lower_left_corner, F32)
    hor = np.array(cam.horizontal, F32)
    ver = np.array(cam.vertical, F32)
    s = (np.arange(W, dtype=F32) / F32(W - 1)).astype(F32)
    t = (np.arange(H, dtype=F32) / F32(H - 1)).astype(F32)
    p = ((llc[None, None, :] + hor[None, None, :] * s[None, :, None]).astype(F32) +
         (ver[None, None, :] * t[:, None, None]).astype(F32)).astype(F32)
    p = (p - o[None, None, :]).astype(F32)
    n2 = ((p[..., 0] * p[..., 0] + p[..., 1] * p[..., 1]).astype(F32) + p[..., 2] * p[..., 2]).astype(F32)
    n = np.sqrt(n2).astype(F32)
    return (p / n[..., None]).astype(F32)  # [H, W, 3]


def camera_struct(eye, look, W, H, skew=None, vfov=60.0):
    cam = esc.Camera.for_image(eye, look, W, H, vfov=vfov).c
    if skew is not None:  # an image plane that is not a rectangle: the C ABI takes any four vectors
        h = np.array(cam.horizontal, F32)
        v = np.array(cam.vertical, F32)
        h2 = (h + F32(skew[0]) * v).astype(F32)
        v2 = (v + F32(skew[1]) * h).astype(F32)
        for k in range(3):
            cam.horizontal[k] = h2[k]
            cam.vertical[k] = v2[k]
    return cam


CAMERAS = [
    ("synthetic view", (0, 3, 6), (0, 2, -8), None, 60.0),
    ("cornell view", (0, 1, 3), (0, 1, 0), None, 60.0),
    ("far from the world origin", (1000.5, 2003, -3006), (1000.5, 2002, -3020), None, 60.0),
    ("skewed image plane", (1, 2, 5), (0.5, 1, -4), (0.3, -0.2), 60.0),
    ("narrow lens", (0, 3, 6), (0, 2, -8), None, 5.0),
    ("wide lens", (0, 3, 6), (0, 2, -8), None, 140.0),
]


@pytest.mark.parametrize("name,eye,look,skew,vfov", CAMERAS, ids=[c[0] for c in CAMERAS])
@pytest.mark.parametrize("W,H", [(320, 180), (129, 65)])
def test_sphere_rectangle_holds_every_pixel_whose_line_passes_the_sphere(name, eye, look, skew, vfov, W, H):
    lib = _capi.load()
    cam = camera_struct(eye, look, W, H, skew, vfov)
    d = ref_dirs(cam, W, H).astype(np.float64).reshape(-1, 3)
    d2 = (d * d).sum(1)
    o = np.array(cam.origin, np.float64)
    fwd = np.array(look, np.float64) - np.array(eye, np.float64)
    dist = np.linalg.norm(fwd)
    fwd /= dist
    rng = np.random.default_rng(zlib.crc32(name.encode()) + W)
    rect = (C.c_int32 * 4)()
    seen = {0: 0, 1: 0, 2: 0}
    checked = 0
    for i in range(1500):
        kind = i % 6
        depth = dist * rng.uniform(0.05, 3.0)
        if kind == 4:
            depth = -depth  # behind the camera: its lines still cross the image
        spread = np.tan(np.radians(vfov) / 2) * (1.2 if kind != 3 else 3.0)
        lateral = rng.normal(size=3) * abs(depth) * min(spread, 3.0)
        c = (o + fwd * depth + lateral).astype(F32)
        R = {0: rng.uniform(0.01, 1.0), 1: 10.0 ** rng.uniform(-6, -2), 2: abs(depth) * rng.uniform(0.3, 0.999),
             3: rng.uniform(0.05, 2.0), 4: rng.uniform(0.05, 1.0), 5: abs(depth) * rng.uniform(0.9, 1.5)}[kind]
        st = lib.esc_tile_rect(C.byref(cam), W, H, c.ctypes.data_as(C.POINTER(C.c_float)), float(R), rect)
        assert st in (0, 1, 2)
        seen[st] += 1
        if st == 0:
            continue  # unbounded: the group is tested by every tile
        cc = c.astype(np.float64) - o
        # C~ = o - fl(o - C) is within 1.01u |c|_1 of C: the binning kernels add that to R themselves
        # (they pass R + 2^-22 |c|_1); here the line's distance from C itself is what is tested
        cd = d @ cc
        dist2 = (cc * cc).sum() - cd * cd / d2
        inside = (dist2 <= R * R).reshape(H, W)
        if not inside.any():
            continue
        hh, ww = np.nonzero(inside)
        checked += 1
        if st == 2:
            raise AssertionError(f"{name}: sphere {c} R={R} declared off screen but covers {len(hh)} pixels")
        assert rect[0] <= ww.min() and ww.max() <= rect[1] and rect[2] <= hh.min() and hh.max() <= rect[3], \
            (name, c, R, list(rect), ww.min(), ww.max(), hh.min(), hh.max())
    assert checked > 150 and seen[1] > 250, (checked, seen)


def test_sphere_rectangle_is_not_much_larger_than_needed():
    """a sanity check of the other direction: for spheres well inside the view the rectangle is the
    tight box of the covered pixels grown by a few pixels, not the whole screen"""
    lib = _capi.load()
    W, H = 640, 360
    cam = camera_struct((0, 3, 6), (0, 2, -8), W, H)
    d = ref_dirs(cam, W, H).astype(np.float64).reshape(-1, 3)
    d2 = (d * d).sum(1)
    o = np.array(cam.origin, np.float64)
    rng = np.random.default_rng(5)
    rect = (C.c_int32 * 4)()
    n = 0
    for _ in range(300):
        c = np.array([rng.uniform(-4, 4), rng.uniform(0.5, 4), rng.uniform(-16, -4)], F32)
        R = rng.uniform(0.1, 1.0)
        st = lib.esc_tile_rect(C.byref(cam), W, H, c.ctypes.data_as(C.POINTER(C.c_float)), float(R), rect)
        cc = c.astype(np.float64) - o
        cd = d @ cc
        inside = ((cc * cc).sum() - cd * cd / d2 <= R * R).reshape(H, W)
        if st != 1 or not inside.any():
            continue
        hh, ww = np.nonzero(inside)
        if ww.min() == 0 or hh.min() == 0 or ww.max() == W - 1 or hh.max() == H - 1:
            continue
        assert rect[0] >= ww.min() - 3 and rect[1] <= ww.max() + 3, (list(rect), ww.min(), ww.max())
        assert rect[2] >= hh.min() - 3 and rect[3] <= hh.max() + 3, (list(rect), hh.min(), hh.max())
        n += 1
    assert n > 100


@pytest.mark.parametrize("name,eye,look,skew,vfov", CAMERAS, ids=[c[0] for c in CAMERAS])
def test_tile_band_holds_every_nearly_parallel_ray(name, eye, look, skew, vfov):
    """(E_t): a triangle the camera is nearly in the plane of can be accepted by rays with |d . n| <= kp
    only; every tile that holds such a ray must be reported, and far fewer than all tiles are."""
    lib = _capi.load()
    W, H = 200, 90  # the last tile column is partly outside the image
    cam = camera_struct(eye, look, W, H, skew, vfov)
    d = ref_dirs(cam, W, H).astype(np.float64)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    tiles_x, tile_rows = (W + 31) // 32, (H + 3) // 4
    fwd = np.array(look, np.float64) - np.array(eye, np.float64)
    fwd /= np.linalg.norm(fwd)
    reported = needed = 0
    for i in range(300):
        # normals nearly perpendicular to the view direction: their bands cross the image
        n = np.cross(fwd, rng.normal(size=3)) + rng.normal(size=3) * (0.3 if i % 3 else 0.02)
        n = (n / np.linalg.norm(n)).astype(F32)
        kp = float(10.0 ** rng.uniform(-6, -1.5))
        dn = np.abs(d @ n.astype(np.float64))  # [H, W]
        for ty in range(tile_rows):
            for tx in range(tiles_x):
                hit = lib.esc_tile_band(C.byref(cam), W, H, tx, 4 * ty, n.ctypes.data_as(C.POINTER(C.c_float)), kp)
                blk = dn[4 * ty:4 * ty + 4, 32 * tx:32 * tx + 32]
                need = blk.size > 0 and bool((blk <= kp).any())
                assert hit in (0, 1)
                assert hit or not need, (name, n, kp, tx, ty, float(blk.min()))
                reported += hit
                needed += need
    assert needed > 200 and reported < 6 * needed + 300 * 30, (needed, reported)
