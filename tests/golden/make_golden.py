#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Runs ONLY in the authoring container: it needs
/root/reference and oracle/_ref/libref_pieces.so (the reference's own vec.h, camera.h,
ray_triangle.h and sceneloader.cpp compiled untouched by oracle/Makefile).

Outputs are DATA (inputs + the reference's outputs), never reference source:
  ref_pieces.npz      seeded random inputs and what the reference's vec / camera /
                      intersect_triangle code returned for them
  loader_<name>.npz   what model::loadobj returned for each bundled OBJ (geometry arrays,
                      material, light list) -- the bundled models are CC-BY / public-domain data
  frames.npz          fp32 frames rendered by OUR oracle (rt_oracle.c) at <= 96x72; these are
                      regression fixtures for the oracle itself, not reference output
"""
import ctypes as C
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol  # noqa: E402

REF_MODELS = "/root/reference/src/models"


def gen_triangle_cases(rng, n):
    """rays aimed at (or near) random triangles, so every branch of ray_triangle.h:23-49 fires"""
    v0 = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    v1 = (v0 + rng.uniform(-2, 2, (n, 3))).astype(np.float32)
    v2 = (v0 + rng.uniform(-2, 2, (n, 3))).astype(np.float32)
    a = rng.uniform(-0.2, 1.2, n)
    b = rng.uniform(-0.2, 1.2, n)
    target = v0 + a[:, None] * (v1 - v0) + b[:, None] * (v2 - v0)
    orig = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    d = target - orig
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    flip = rng.random(n) < 0.1  # triangle behind the origin
    d[flip] *= -1
    dirs = d.astype(np.float32)
    dist = np.linalg.norm(target - orig, axis=1)
    tb = np.where(rng.random(n) < 0.5, np.float32(np.finfo(np.float32).max),
                  (dist * rng.uniform(0.5, 1.5, n))).astype(np.float32)
    k = n // 20  # exact edge / vertex / degenerate / parallel cases
    target[:k] = v0[:k]  # through vertex 0 (u = v = 0)
    target[k:2 * k] = (v0[k:2 * k] + 0.5 * (v1[k:2 * k] - v0[k:2 * k]))  # on edge v = 0
    for sl in (slice(0, k), slice(k, 2 * k)):
        dd = target[sl] - orig[sl]
        dirs[sl] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    v2[2 * k:3 * k] = v1[2 * k:3 * k]  # degenerate triangle
    e = (v1[3 * k:4 * k] - v0[3 * k:4 * k])
    dirs[3 * k:4 * k] = (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)  # in-plane
    return orig, dirs, v0, v1, v2, tb


def ref_pieces(path):
    r = ol.ref()
    rng = np.random.default_rng(20211004)
    n = 20000
    orig, dirs, v0, v1, v2, tb = gen_triangle_cases(rng, n)
    hit = np.zeros(n, np.int32)
    tuv = np.zeros((n, 3), np.float32)
    for i in range(n):
        t, u, v = C.c_float(tb[i]), C.c_float(-7.0), C.c_float(-7.0)
        hit[i] = r.ref_intersect_triangle(ol.fp(orig[i]), ol.fp(dirs[i]), ol.fp(v0[i]),
                                          ol.fp(v1[i]), ol.fp(v2[i]), C.byref(t), C.byref(u),
                                          C.byref(v))
        tuv[i] = (t.value, u.value, v.value)
    # vec.h
    m = 2000
    a = rng.normal(0, 3, (m, 3)).astype(np.float32)
    b = rng.normal(0, 3, (m, 3)).astype(np.float32)
    s = rng.uniform(0.1, 5, m).astype(np.float32)
    vdot = np.zeros(m, np.float32)
    vlen = np.zeros(m, np.float32)
    vcross = np.zeros((m, 3), np.float32)
    vnorm = np.zeros((m, 3), np.float32)
    vdiv = np.zeros((m, 3), np.float32)
    for i in range(m):
        vdot[i] = r.ref_dot(ol.fp(a[i]), ol.fp(b[i]))
        vlen[i] = r.ref_length(ol.fp(a[i]))
        r.ref_cross(ol.fp(a[i]), ol.fp(b[i]), ol.fp(vcross[i]))
        r.ref_normalize(ol.fp(a[i]), ol.fp(vnorm[i]))
        r.ref_div(ol.fp(a[i]), C.c_float(s[i]), ol.fp(vdiv[i]))
    # camera.h
    c = 200
    eye = rng.uniform(-5, 5, (c, 3)).astype(np.float32)
    look = rng.uniform(-5, 5, (c, 3)).astype(np.float32)
    vfov = rng.uniform(20, 100, c).astype(np.float32)
    aspect = rng.uniform(0.5, 2.5, c).astype(np.float32)
    st = rng.uniform(0, 1, (c, 2)).astype(np.float32)
    cam12 = np.zeros((c, 12), np.float32)
    rays = np.zeros((c, 3), np.float32)
    up = np.array([0, 1, 0], np.float32)
    # the reference's defaults first (main.cpp:426-427,548-550)
    eye[0], look[0], vfov[0], aspect[0] = (0, 1, 3), (0, 1, 0), 60, np.float32(1024) / np.float32(768)
    for i in range(c):
        r.ref_camera(ol.fp(eye[i]), ol.fp(look[i]), ol.fp(up), C.c_float(vfov[i]),
                     C.c_float(aspect[i]), ol.fp(cam12[i]))
        r.ref_get_ray(ol.fp(eye[i]), ol.fp(look[i]), ol.fp(up), C.c_float(vfov[i]),
                      C.c_float(aspect[i]), C.c_float(st[i, 0]), C.c_float(st[i, 1]),
                      ol.fp(rays[i]))
    np.savez_compressed(path, tri_orig=orig, tri_dir=dirs, tri_v0=v0, tri_v1=v1, tri_v2=v2,
                        tri_tbound=tb, tri_hit=hit, tri_tuv=tuv, vec_a=a, vec_b=b, vec_s=s,
                        vec_dot=vdot, vec_len=vlen, vec_cross=vcross, vec_norm=vnorm,
                        vec_div=vdiv, cam_eye=eye, cam_look=look, cam_vfov=vfov,
                        cam_aspect=aspect, cam_st=st, cam_vectors=cam12, cam_rays=rays)
    print("ref_pieces: hits", int(hit.sum()), "of", n)


def ref_load(path):
    r = ol.ref()
    h = r.ref_loadobj(path.encode())
    err = r.ref_scene_error(h).decode()
    if err:
        r.ref_scene_free(h)
        return None
    out = {}
    ng = r.ref_scene_n_geometry(h)
    for g in range(ng):
        cnt = (C.c_int * 3)()
        r.ref_geom_counts(h, g, cnt)
        v = np.zeros((cnt[0], 3), np.float32)
        n = np.zeros((max(cnt[1], 1), 3), np.float32)
        f = np.zeros((cnt[2], 3), np.uint32)
        m = np.zeros(13, np.float32)
        r.ref_geom_copy(h, g, ol.fp(v), ol.fp(n), f.ctypes.data_as(C.POINTER(C.c_uint32)), ol.fp(m))
        out[f"g{g}_vertex"] = v
        out[f"g{g}_normals"] = n[:cnt[1]]
        out[f"g{g}_face_index"] = f
        out[f"g{g}_material"] = m
    out["n_geometry"] = np.array(ng)
    out["light_sources"] = np.array([r.ref_scene_light(h, i)
                                     for i in range(r.ref_scene_n_lights(h))], np.int32)
    r.ref_scene_free(h)
    return out


def loader_dumps():
    throws = []
    for p in sorted(glob.glob(REF_MODELS + "/**/*.obj", recursive=True)) + \
            sorted(glob.glob(os.path.join(HERE, "scenes", "*.obj"))):
        name = os.path.splitext(os.path.basename(p))[0]
        d = ref_load(p)
        if d is None:
            throws.append(name)
            continue
        np.savez_compressed(os.path.join(HERE, f"loader_{name}.npz"), **d)
        print("loader dump", name, int(d["n_geometry"]), "geometries")
    np.savez_compressed(os.path.join(HERE, "loader_throws.npz"), names=np.array(throws))
    print("reference loader throws on:", throws)


def frames():
    out = {}
    for name, eye in (("one", (0, 1, 3)), ("two", (0, 1, 3)), ("CornellBox-Original", (0, 1, 2))):
        d = ol.load_dump(os.path.join(HERE, f"loader_{name}.npz"))
        out[name] = ol.oracle_render(d, eye, (0, 1, 0), 96, 72)
    d = ol.load_dump(os.path.join(HERE, "loader_CornellBox-Original.npz"))
    out["CornellBox-Original_face1"] = ol.oracle_render(d, (0, 1, 2), (0, 1, 0), 96, 72, fixed_face=1)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)
    print("frames:", {k: float(v.sum()) for k, v in out.items()})


if __name__ == "__main__":
    if not ol.have_ref():
        sys.exit("oracle/_ref/libref_pieces.so missing: run `make -C oracle` where /root/reference exists")
    ref_pieces(os.path.join(HERE, "ref_pieces.npz"))
    loader_dumps()
    frames()
