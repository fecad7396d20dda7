"""ctypes access to the test-only CPU checker (oracle/liboracle.so) and, when it has been
built in the authoring container, to the reference's own code (oracle/_ref/libref_pieces.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
ORACLE_FAST_SO = os.path.join(ORACLE_DIR, "liboracle_fast.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_pieces.so")

ORC_FACE_FIXED, ORC_FACE_HASH = 0, 1
ORC_QUIRK_S1, ORC_QUIRK_S3, ORC_QUIRK_ALL = 1, 2, 3

_F = C.POINTER(C.c_float)


class orc_material(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3),
                ("ke", C.c_float * 3), ("Ns", C.c_float)]


class orc_geometry(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("vertex", _F), ("n_normals", C.c_int32),
                ("normals", _F), ("n_faces", C.c_int32), ("face_index", C.POINTER(C.c_uint32)),
                ("material", orc_material)]


class orc_scene(C.Structure):
    _fields_ = [("n_geometry", C.c_int32), ("geometry", C.POINTER(orc_geometry)),
                ("n_lights", C.c_int32), ("light_sources", C.POINTER(C.c_int32)),
                ("n_spheres", C.c_int32), ("spheres", _F),
                ("sphere_material", C.POINTER(C.c_int32)),
                ("sphere_materials", C.POINTER(orc_material))]


class orc_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3)]


class orc_options(C.Structure):
    _fields_ = [("shadows", C.c_int32), ("face_mode", C.c_int32), ("fixed_face", C.c_int32),
                ("seed", C.c_uint64), ("quirks", C.c_int32)]


class orc_counters(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("hit_pixels", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("anyhit_tests", C.c_uint64)]


_oracle = None
_ref = None


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.orc_dot.restype = C.c_float
        lib.orc_dot.argtypes = [_F, _F]
        lib.orc_cross.argtypes = [_F, _F, _F]
        lib.orc_normalize.argtypes = [_F, _F]
        lib.orc_length.restype = C.c_float
        lib.orc_length.argtypes = [_F]
        lib.orc_camera_init.argtypes = [C.POINTER(orc_camera), _F, _F, _F, C.c_float, C.c_float]
        lib.orc_camera_get_ray.argtypes = [C.POINTER(orc_camera), C.c_float, C.c_float, _F]
        lib.orc_intersect_triangle.restype = C.c_int
        lib.orc_intersect_triangle.argtypes = [_F, _F, _F, _F, _F, _F, _F, _F]
        lib.orc_intersect_sphere.restype = C.c_int
        lib.orc_intersect_sphere.argtypes = [_F, _F, _F, _F]
        lib.orc_face_hash.restype = C.c_uint32
        lib.orc_face_hash.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.orc_render.argtypes = [C.POINTER(orc_scene), C.POINTER(orc_camera), C.c_int32,
                                   C.c_int32, C.c_int32, C.c_int32, C.POINTER(orc_options), _F,
                                   C.POINTER(orc_counters), C.c_int32]
        lib.orc_render_row_list.argtypes = [C.POINTER(orc_scene), C.POINTER(orc_camera), C.c_int32,
                                            C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                            C.POINTER(orc_options), _F, C.POINTER(orc_counters),
                                            C.c_int32]
        lib.orc_quantise.argtypes = [_F, C.c_int64, C.POINTER(C.c_uint8)]
        lib.orc_write_ppm.restype = C.c_int
        lib.orc_write_ppm.argtypes = [C.c_char_p, _F, C.c_int32, C.c_int32]
        _oracle = lib
    return _oracle


_oracle_fast = None


def have_avx2():
    try:
        with open("/proc/cpuinfo") as f:
            return " avx2" in f.read()
    except OSError:
        return False


def oracle_fast():
    """rt_oracle_fast.c: the same restatement eight pixels at a time (AVX2).  None when the
    host CPU has no AVX2."""
    global _oracle_fast
    if _oracle_fast is None:
        if not have_avx2():
            return None
        if not os.path.exists(ORACLE_FAST_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_FAST_SO)
        lib.orc_fast_render_row_list.argtypes = [
            C.POINTER(orc_scene), C.POINTER(orc_camera), C.c_int32, C.c_int32,
            C.POINTER(C.c_int32), C.c_int32, C.POINTER(orc_options), _F, C.POINTER(orc_counters),
            C.c_int32]
        _oracle_fast = lib
    return _oracle_fast


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The reference's own vec.h / camera.h / ray_triangle.h / sceneloader.cpp."""
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ref_dot.restype = C.c_float
        lib.ref_dot.argtypes = [_F, _F]
        lib.ref_length.restype = C.c_float
        lib.ref_length.argtypes = [_F]
        for n in ("ref_cross", "ref_add", "ref_sub"):
            getattr(lib, n).argtypes = [_F, _F, _F]
        lib.ref_normalize.argtypes = [_F, _F]
        lib.ref_scale.argtypes = [_F, C.c_float, _F]
        lib.ref_div.argtypes = [_F, C.c_float, _F]
        lib.ref_camera.argtypes = [_F, _F, _F, C.c_float, C.c_float, _F]
        lib.ref_get_ray.argtypes = [_F, _F, _F, C.c_float, C.c_float, C.c_float, C.c_float, _F]
        lib.ref_intersect_triangle.restype = C.c_int
        lib.ref_intersect_triangle.argtypes = [_F, _F, _F, _F, _F, _F, _F, _F]
        lib.ref_loadobj.restype = C.c_void_p
        lib.ref_loadobj.argtypes = [C.c_char_p]
        lib.ref_scene_free.argtypes = [C.c_void_p]
        lib.ref_scene_error.restype = C.c_char_p
        lib.ref_scene_error.argtypes = [C.c_void_p]
        lib.ref_scene_n_geometry.argtypes = [C.c_void_p]
        lib.ref_scene_n_lights.argtypes = [C.c_void_p]
        lib.ref_scene_light.argtypes = [C.c_void_p, C.c_int]
        lib.ref_geom_counts.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        lib.ref_geom_copy.argtypes = [C.c_void_p, C.c_int, _F, _F, C.POINTER(C.c_uint32), _F]
        _ref = lib
    return _ref


def fp(a):
    return a.ctypes.data_as(_F)


def f3(x):
    return np.ascontiguousarray(x, dtype=np.float32).reshape(3)


# --------------------------------------------------------------------------------------
# scene description shared by the oracle and the product: a plain dict
#   {"geometry": [{"vertex": (N,3) f32, "normals": (M,3) f32, "face_index": (F,3) u32,
#                  "material": (13,) f32}, ...],
#    "light_sources": [geom ids], "spheres": (K,4) f32, "sphere_materials": (K,13) f32}
# --------------------------------------------------------------------------------------
def material13(ka=(0, 0, 0), kd=(0, 0, 0), ks=(0, 0, 0), ke=(0, 0, 0), Ns=10.0):
    return np.array(list(ka) + list(kd) + list(ks) + list(ke) + [Ns], np.float32)


def scene_dict(geometry, spheres=None, sphere_materials=None):
    lights = []
    for i, g in enumerate(geometry):
        g["vertex"] = np.ascontiguousarray(g["vertex"], np.float32).reshape(-1, 3)
        g["face_index"] = np.ascontiguousarray(g["face_index"], np.uint32).reshape(-1, 3)
        n = g.get("normals")
        g["normals"] = (np.zeros((0, 3), np.float32) if n is None or len(n) == 0
                        else np.ascontiguousarray(n, np.float32).reshape(-1, 3))
        g["material"] = np.ascontiguousarray(g["material"], np.float32).reshape(13)
        ke = g["material"][9:12]
        s = np.float32(0)
        for k in range(3):  # sceneloader.cpp:63-64 in vec.h:95-101 order
            s = np.float32(s + np.float32(ke[k] * ke[k]))
        if s > 0:
            lights.append(i)
    sp = np.zeros((0, 4), np.float32) if spheres is None else \
        np.ascontiguousarray(spheres, np.float32).reshape(-1, 4)
    sm = np.zeros((0, 13), np.float32) if sphere_materials is None else \
        np.ascontiguousarray(sphere_materials, np.float32).reshape(-1, 13)
    return {"geometry": geometry, "light_sources": lights, "spheres": sp, "sphere_materials": sm}


def scene_from_product(scene):
    """esctp1raytracer_amd.Scene -> dict, through the C ABI's introspection calls."""
    info = scene.info()
    geometry = [scene.geometry(i) for i in range(info["n_geometry"])]
    sp, sm = scene.spheres()
    d = scene_dict(geometry, sp, sm)
    assert d["light_sources"] == list(scene.light_sources()), "light list mismatch"
    return d


def scene_to_product(d):
    import esctp1raytracer_amd as esc
    sc = esc.Scene()
    for g in d["geometry"]:
        sc.add_geometry(g["vertex"], g["face_index"], g["material"], g["normals"])
    if len(d["spheres"]):
        sc.add_spheres(d["spheres"], d["sphere_materials"])
    return sc


class OracleScene:
    """Keeps the numpy buffers alive behind an orc_scene."""

    def __init__(self, d):
        self.keep = []
        ng = len(d["geometry"])
        self.geoms = (orc_geometry * max(ng, 1))()
        for i, g in enumerate(d["geometry"]):
            og = self.geoms[i]
            v, n, f = g["vertex"], g["normals"], g["face_index"]
            self.keep += [v, n, f]
            og.n_vertices = v.shape[0]
            og.vertex = fp(v)
            og.n_normals = n.shape[0]
            og.normals = fp(n) if n.shape[0] else None
            og.n_faces = f.shape[0]
            og.face_index = f.ctypes.data_as(C.POINTER(C.c_uint32))
            og.material = _mat(g["material"])
        self.lights = np.array(d["light_sources"] or [0], np.int32)
        sp = d["spheres"]
        self.sph = sp
        self.sph_mat_idx = np.arange(max(len(sp), 1), dtype=np.int32)
        self.sph_mats = (orc_material * max(len(sp), 1))()
        for k in range(len(sp)):
            self.sph_mats[k] = _mat(d["sphere_materials"][k])
        s = orc_scene()
        s.n_geometry = ng
        s.geometry = self.geoms
        s.n_lights = len(d["light_sources"])
        s.light_sources = self.lights.ctypes.data_as(C.POINTER(C.c_int32))
        s.n_spheres = len(sp)
        s.spheres = fp(sp) if len(sp) else None
        s.sphere_material = self.sph_mat_idx.ctypes.data_as(C.POINTER(C.c_int32))
        s.sphere_materials = self.sph_mats
        self.c = s


def _mat(m13):
    m = orc_material()
    for i in range(3):
        m.ka[i] = m13[i]
        m.kd[i] = m13[3 + i]
        m.ks[i] = m13[6 + i]
        m.ke[i] = m13[9 + i]
    m.Ns = m13[12]
    return m


def oracle_camera(lookfrom, lookat, W, H, vup=(0, 1, 0), vfov=60.0):
    cam = orc_camera()
    aspect = np.float32(W) / np.float32(H)  # main.cpp:548
    oracle().orc_camera_init(C.byref(cam), fp(f3(lookfrom)), fp(f3(lookat)), fp(f3(vup)),
                             C.c_float(vfov), C.c_float(aspect))
    return cam


def oracle_render(d, lookfrom, lookat, W, H, *, shadows=True, face_mode=ORC_FACE_FIXED,
                  fixed_face=0, seed=0, quirks=ORC_QUIRK_ALL, rows=None, threads=1,
                  return_counters=False, vfov=60.0):
    """fp32 (H, W, 3) image, h = 0 bottom row; rows outside `rows` stay zero."""
    osc = d if isinstance(d, OracleScene) else OracleScene(d)
    cam = oracle_camera(lookfrom, lookat, W, H, vfov=vfov)
    o = orc_options(1 if shadows else 0, face_mode, fixed_face, seed, quirks)
    img = np.zeros((H, W, 3), np.float32)
    cnt = orc_counters()
    r0, r1 = rows if rows is not None else (0, H)
    oracle().orc_render(C.byref(osc.c), C.byref(cam), W, H, r0, r1, C.byref(o), fp(img),
                        C.byref(cnt), threads)
    if return_counters:
        return img, {"primary_rays": cnt.primary_rays, "hit_pixels": cnt.hit_pixels,
                     "shadow_rays": cnt.shadow_rays, "anyhit_tests": cnt.anyhit_tests}
    return img


def oracle_render_rows(d, lookfrom, lookat, W, H, rows, *, shadows=True,
                       face_mode=ORC_FACE_FIXED, fixed_face=0, seed=0, threads=1, fast=False):
    """rows: list of image rows -> (len(rows), W, 3) fp32 + counters.  fast=True runs the
    eight-pixel packet version (rt_oracle_fast.c; falls back to the scalar one without AVX2)."""
    osc = d if isinstance(d, OracleScene) else OracleScene(d)
    cam = oracle_camera(lookfrom, lookat, W, H)
    o = orc_options(1 if shadows else 0, face_mode, fixed_face, seed, ORC_QUIRK_ALL)
    rr = np.ascontiguousarray(rows, np.int32)
    img = np.zeros((len(rr), W, 3), np.float32)
    cnt = orc_counters()
    lib_fast = oracle_fast() if fast else None
    fn = lib_fast.orc_fast_render_row_list if lib_fast is not None else \
        oracle().orc_render_row_list
    fn(C.byref(osc.c), C.byref(cam), W, H, rr.ctypes.data_as(C.POINTER(C.c_int32)), len(rr),
       C.byref(o), fp(img), C.byref(cnt), threads)
    return img, {"primary_rays": cnt.primary_rays, "hit_pixels": cnt.hit_pixels,
                 "shadow_rays": cnt.shadow_rays, "anyhit_tests": cnt.anyhit_tests}


def oracle_quantise(img):
    out = np.zeros(img.shape, np.uint8)
    a = np.ascontiguousarray(img, np.float32)
    oracle().orc_quantise(fp(a), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


# --------------------------------------------------------------------------------------
# SURVEY.md Appendix B fixture scenes, as the reference's loader de-indexes them
# --------------------------------------------------------------------------------------
WHITE = material13(ka=(0.725, 0.71, 0.68), kd=(0.725, 0.71, 0.68))
RED = material13(ka=(0.63, 0.065, 0.05), kd=(0.63, 0.065, 0.05))
BLUE = material13(ka=(0.1, 0.2, 0.7), kd=(0.1, 0.2, 0.7))
LIGHT_A = material13(ka=(0.78, 0.78, 0.78), kd=(0.78, 0.78, 0.78), ke=(17, 12, 4))
LIGHT_B = material13(ka=(0.5, 0.5, 0.5), kd=(0.5, 0.5, 0.5), ke=(3, 5, 9))


def _tris(points, faces, normals=None, nfaces=None):
    """de-index like sceneloader.cpp:73-98: three fresh vertices per face."""
    v = np.array([points[i] for f in faces for i in f], np.float32)
    fi = np.arange(len(faces) * 3, dtype=np.uint32).reshape(-1, 3)
    n = None
    if normals is not None:
        raw = np.array([normals[i] for f in nfaces for i in f], np.float32)
        n = np.zeros_like(raw)
        for k in range(len(raw)):  # sceneloader.cpp:88 normalize(n) with vec.h order
            s = np.float32(0)
            for c in range(3):
                s = np.float32(s + np.float32(raw[k, c] * raw[k, c]))
            n[k] = raw[k] / np.sqrt(s, dtype=np.float32)
    return v, fi, n


def scene_one():
    p = [(-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2), (-0.5, 0.3, 0.2), (0.6, 0.4, -0.1),
         (0.1, 1.2, -0.3), (-0.3, 1.98, 0.3), (0.3, 1.98, 0.3), (0.0, 1.98, -0.3)]
    g = []
    for faces, m in (([(0, 1, 2), (0, 2, 3)], WHITE), ([(4, 5, 6)], RED), ([(7, 9, 8)], LIGHT_A)):
        v, fi, _ = _tris(p, faces)
        g.append({"vertex": v, "face_index": fi, "material": m})
    return scene_dict(g)


def scene_two():
    p = [(-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2), (-0.6, 0.2, 0.1), (0.5, 0.3, -0.2),
         (0.0, 1.1, -0.4), (0.9, 0.9, -0.9), (-0.3, 1.98, 0.3), (0.3, 1.98, 0.3),
         (0.0, 1.98, -0.3), (1.5, 1.2, 1.0), (1.5, 1.8, 1.0), (1.5, 1.5, 0.4)]
    vn = [(-0.3, 0.4, 0.86), (0.3, 0.5, 0.81), (0.0, 0.7, 0.71), (0.5, 0.5, 0.70)]
    g = []
    v, fi, _ = _tris(p, [(0, 1, 2), (0, 2, 3)])
    g.append({"vertex": v, "face_index": fi, "material": WHITE})
    v, fi, n = _tris(p, [(4, 5, 6), (5, 7, 6)], vn, [(0, 1, 2), (1, 3, 2)])
    g.append({"vertex": v, "face_index": fi, "material": BLUE, "normals": n})
    v, fi, _ = _tris(p, [(8, 10, 9)])
    g.append({"vertex": v, "face_index": fi, "material": LIGHT_A})
    v, fi, _ = _tris(p, [(11, 12, 13)])
    g.append({"vertex": v, "face_index": fi, "material": LIGHT_B})
    return scene_dict(g)


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def load_dump(name_or_path):
    """tests/golden/loader_<name>.npz (what the reference's model::loadobj returned) -> dict"""
    path = name_or_path if os.path.exists(name_or_path) else \
        os.path.join(GOLDEN_DIR, f"loader_{name_or_path}.npz")
    z = np.load(path)
    geometry = [{"vertex": z[f"g{g}_vertex"], "normals": z[f"g{g}_normals"],
                 "face_index": z[f"g{g}_face_index"], "material": z[f"g{g}_material"]}
                for g in range(int(z["n_geometry"]))]
    d = scene_dict(geometry)
    assert d["light_sources"] == list(z["light_sources"])
    return d


# --------------------------------------------------------------------------------------
# SURVEY.md 8(d) "triangle-parity companions": a sphere config re-expressed with the
# reference's ONLY primitive (ray_triangle.h:7-57) -- every sphere becomes a tessellated
# icosahedron, one geometry per sphere, rendered through the reference-pinned triangle path.
# --------------------------------------------------------------------------------------
def _icosphere(subdiv):
    """unit icosphere: (V,3) float64 vertices, (F,3) int faces (20 * 4**subdiv faces)"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t),
         (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
         (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9),
         (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v), np.array(f, np.int64)


def icosphere_twin(d, subdiv, smooth_normals=False):
    """scene dict with spheres -> scene dict with triangles only: sphere k (centre c, radius r,
    material m) becomes geometry n_geometry + k, an icosphere of 20 * 4**subdiv faces with
    de-indexed vertices (3 per face, like sceneloader.cpp:73-98) and optionally per-vertex
    normals (exercises quirk S1 on every one of them)."""
    uv, uf = _icosphere(subdiv)
    geometry = list(d["geometry"])
    corners = uv[uf.reshape(-1)]  # (F*3, 3) unit directions, face by face
    fi = np.arange(len(corners), dtype=np.uint32).reshape(-1, 3)
    for k in range(len(d["spheres"])):
        cx, cy, cz, r = (float(x) for x in d["spheres"][k])
        vert = (corners * r + np.array([cx, cy, cz])).astype(np.float32)
        g = {"vertex": vert, "face_index": fi.copy(), "material": d["sphere_materials"][k].copy()}
        if smooth_normals:
            n = corners.astype(np.float32)
            s = np.sqrt((n * n).sum(axis=1, dtype=np.float32), dtype=np.float32)
            g["normals"] = (n / s[:, None]).astype(np.float32)
        geometry.append(g)
    return scene_dict(geometry)
