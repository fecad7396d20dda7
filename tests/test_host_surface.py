"""CPU-side tests of the drop-in boundary: the C ABI loads and exports every declared symbol,
the host half (scene model, loader, camera, flatten, PPM) behaves like the reference's, and
the product refuses to render without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = ol.ROOT


@pytest.fixture(scope="module")
def esc():
    import esctp1raytracer_amd as m
    return m


def test_library_exports_every_declared_symbol(esc):
    from esctp1raytracer_amd import _capi
    header = open(os.path.join(ROOT, "include", "esctp1_rt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(esc_[a-z0-9_]+|trace)\s*\(", header))
    declared -= {"esc_scene", "esc_context", "esc_flat_scene", "esc_multi"}
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = _capi.load()
    for name in declared:
        assert getattr(lib, name) is not None
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True,
                        text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    assert declared <= exported


@pytest.mark.parametrize("n_tri,n_sph", [(3, 10000), (0, 10000), (100353, 0), (200003, 0), (3, 2047),
                                         (65, 1), (64, 9), (1, 0), (0, 1), (0, 0), (7, 7),
                                         (1000, 100001), (12345, 6789)])
def test_queue_schedule_covers_every_primitive_once_in_order(n_tri, n_sph):
    """esc_queue_schedule: the segments the queue form sweeps.  occlusion() (main.cpp:314-329)
    meets triangles in index order, then spheres; every primitive must be in exactly one segment,
    segments in ascending order, triangle segments starting on even indices (pair-interleaved
    filter records), sphere segments in whole pair records starting on multiples of 4."""
    from esctp1raytracer_amd import _capi
    lib = _capi.load()
    segs = (C.c_int32 * (4 * 64))()
    n = lib.esc_queue_schedule(n_tri, n_sph, segs, 64)
    assert 0 <= n <= 48
    n_rec = (n_sph + 1) // 2
    tri_next, rec_next = 0, 0
    for i in range(n):
        t0, tn, r0, rn = segs[4 * i:4 * i + 4]
        assert tn >= 0 and rn >= 0 and tn + rn > 0
        if tn:
            assert t0 == tri_next and t0 % 2 == 0
            assert rec_next == 0 or rn == 0  # triangles come before any sphere is swept
            tri_next += tn
        if rn:
            assert tri_next == n_tri, "spheres before the triangles are done"
            assert r0 == rec_next and r0 % 4 == 0
            rec_next += rn
    assert tri_next == n_tri and rec_next == n_rec
    assert lib.esc_queue_schedule(n_tri, n_sph, segs, 0) < 0 or n == 0
    assert lib.esc_queue_schedule(-1, 0, segs, 64) < 0


def test_multi_gpu_entry_without_a_gpu_fails_loudly(esc):
    """esc_multi_create / esc_render_frame_multi_rccl on a host without a GPU: ESC_ERR_NO_DEVICE,
    never a CPU render; the RCCL probe itself must not need a device."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    from esctp1raytracer_amd import _capi
    assert esc.rccl_available() in (True, False)
    with pytest.raises(esc.EscError) as e:
        esc.MultiRenderer(2)
    assert e.value.code == _capi.ESC_ERR_NO_DEVICE
    sc = esc.Scene.synthetic("c2", 10)
    cam = esc.Camera.for_image((0, 3, 6), (0, 2, -8), 64, 48)
    with pytest.raises(esc.EscError) as e:
        esc.render_multi_rccl(sc, cam, 64, 48, 2)
    assert e.value.code == _capi.ESC_ERR_NO_DEVICE
    with pytest.raises(esc.EscError):
        esc.MultiRenderer(0)


def test_struct_layouts_match_ispc_headers():
    """ispc_helpers.h:16-29,52-65: 140-byte triangle, 24-byte light (LP64), 44-byte camera"""
    from esctp1raytracer_amd import _capi
    assert C.sizeof(_capi.ispc_triangle) == 140
    assert C.sizeof(_capi.ispc_light) == 24
    assert C.sizeof(_capi.ispc_cam) == 44
    assert _capi.ispc_triangle.ka.offset == 88 and _capi.ispc_triangle.Ns.offset == 136


def test_library_contains_gfx950_code_object():
    from esctp1raytracer_amd import _capi
    blob = open(_capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_primary" in blob and b"k_shade" in blob


def test_no_gpu_means_error_not_fallback(esc):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(esc.EscError) as e:
        esc.Renderer(0)
    assert e.value.code == -6  # ESC_ERR_NO_DEVICE
    sc = esc.Scene.synthetic("c2")
    cam = esc.Camera.for_image(*esc.synthetic_view(), 64, 48)
    with pytest.raises(esc.EscError):
        esc.render_multi(sc, cam, 64, 48, 2)


def test_bench_and_viewer_fail_loudly_without_a_gpu():
    """no silent CPU path anywhere: bench.py and the viewer exit non-zero with a clear message"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
    viewer = os.path.join(ROOT, "bin", "ESCViewer2021")
    if os.path.exists(viewer):
        v = subprocess.run([viewer, "--scene", "c2", "-w", "64,48"], capture_output=True,
                           text=True, timeout=120)
        assert v.returncode != 0 and "no CPU fallback" in (v.stderr + v.stdout)


def test_product_does_not_link_or_import_the_oracle():
    from esctp1raytracer_amd import _capi
    ldd = subprocess.run(["ldd", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "ref_pieces" not in ldd
    for dirpath, _, files in os.walk(os.path.join(ROOT, "esctp1raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"#\s*include[^\n]*rt_oracle|import\s+oracle_lib|"
                                     r"from\s+oracle_lib|liboracle|libref_pieces|orc_[a-z_]+\(",
                                     txt), f


# ------------------------------------------------------------------ loader (sceneloader.cpp)
LOADER_DUMPS = ["one", "two", "CornellBox-Original", "CornellBox-Mirror", "CornellBox-Sphere",
                "CornellBox-Water", "CornellBox-Empty-CO", "CornellBox-Empty-RG",
                "CornellBox-Empty-White", "CornellBox-Empty-Squashed", "cornell_box", "water"]


def _assert_same_scene(mine, refd):
    assert len(mine["geometry"]) == len(refd["geometry"])
    assert mine["light_sources"] == refd["light_sources"]
    for a, b in zip(mine["geometry"], refd["geometry"]):
        for k in ("vertex", "normals", "face_index", "material"):
            assert a[k].shape == b[k].shape, k
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


@pytest.mark.parametrize("name", ["one", "two"])
def test_loader_on_committed_obj_fixtures(esc, name, golden_dir):
    sc = esc.Scene.load_obj(os.path.join(golden_dir, "scenes", name + ".obj"))
    _assert_same_scene(ol.scene_from_product(sc), ol.load_dump(name))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/models"),
                    reason="bundled models live in the reference tree only")
@pytest.mark.parametrize("name", LOADER_DUMPS[2:])
def test_loader_bit_equal_to_reference_loader_dump(esc, name):
    """every bundled OBJ the reference can load: geometry split (quirk S11), de-indexing,
    tinyobj's digit-by-digit float parse, normalised normals, light list"""
    sub = "" if name in ("cornell_box",) else "cornell/"
    sc = esc.Scene.load_obj(f"/root/reference/src/models/{sub}{name}.obj")
    _assert_same_scene(ol.scene_from_product(sc), ol.load_dump(name))
    if name == "CornellBox-Original":  # S11: short box joins leftWall
        assert [len(g["face_index"]) for g in ol.load_dump(name)["geometry"]] == [2, 2, 2, 2, 14, 12, 2]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/models"), reason="needs bundled models")
def test_loader_rejects_what_the_reference_rejects(esc, golden_dir):
    throws = [str(n) for n in np.load(golden_dir + "/loader_throws.npz")["names"]]
    assert sorted(throws) == ["CornellBox-Glossy", "CornellBox-Glossy-Floor"]  # quirk S10
    for n in throws:
        with pytest.raises(esc.EscError) as e:
            esc.Scene.load_obj(f"/root/reference/src/models/cornell/{n}.obj")
        assert e.value.code == -4


def test_loader_errors(esc, tmp_path):
    with pytest.raises(esc.EscError) as e:
        esc.Scene.load_obj(tmp_path / "nope.obj")
    assert e.value.code == -3
    p = tmp_path / "nomtl.obj"
    p.write_text("mtllib missing.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl x\nf 1 2 3\n")
    with pytest.raises(esc.EscError) as e:  # "WARN: Material file not found" is fatal (S10)
        esc.Scene.load_obj(p)
    assert e.value.code == -4


def test_loader_number_parser_and_polygons(esc, tmp_path):
    """negative indices, a quad (fan triangulation), exponents, v/t/n corners"""
    (tmp_path / "m.mtl").write_text("newmtl a\nKa 0.1 0.2 0.3\nKd 1e-1 2.5E+0 .5\nKe 0 0 0\nNs 3\n")
    (tmp_path / "q.obj").write_text(
        "mtllib m.mtl\nv 0 0 0\nv 1.5 0 0\nv 1.5 2.25 0\nv 0 2.25 1e-3\nvt 0 0\nvn 0 0 2\n"
        "g quad\nusemtl a\nf -4/1/1 -3/1/1 -2/1/1 -1/1/1\n")
    d = ol.scene_from_product(esc.Scene.load_obj(tmp_path / "q.obj"))
    g = d["geometry"][0]
    assert g["face_index"].tolist() == [[0, 1, 2], [3, 4, 5]]
    assert g["vertex"].tolist() == [[0, 0, 0], [1.5, 0, 0], [1.5, 2.25, 0],
                                    [0, 0, 0], [1.5, 2.25, 0], [0, 2.25, np.float32(1e-3)]]
    assert np.allclose(g["normals"], [[0, 0, 1]] * 6)
    assert g["material"].tolist()[:6] == [np.float32(0.1), np.float32(0.2), np.float32(0.3),
                                          np.float32(0.1), 2.5, 0.0]  # ".5" has no leading digit
    assert g["material"][12] == 3.0


# ------------------------------------------------------------------ scene model / camera
def test_scene_build_introspect_roundtrip(esc):
    d = ol.scene_two()
    sc = ol.scene_to_product(d)
    _assert_same_scene(ol.scene_from_product(sc), d)
    assert sc.info() == {"n_geometry": 4, "n_lights": 2, "n_triangles": 6, "n_spheres": 0}
    with pytest.raises(esc.EscError):
        sc.add_geometry([[0, 0, 0]] * 3, [[0, 1, 3]], ol.WHITE)  # face index out of range


def test_scene_rejects_bad_input(esc):
    """out-of-range face indices and non-finite coordinates fail loudly at scene building"""
    m = ol.material13(ka=(.5,) * 3, kd=(.5,) * 3)
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    sc = esc.Scene()
    with pytest.raises(esc.EscError):
        sc.add_geometry(v, np.array([[0, 1, 3]]), m)
    bad = v.copy()
    bad[1, 2] = np.nan
    with pytest.raises(esc.EscError):
        sc.add_geometry(bad, np.array([[0, 1, 2]]), m)
    with pytest.raises(esc.EscError):
        sc.add_spheres(np.array([[0, 0, 0, np.inf]], np.float32), m[None, :])
    assert sc.info()["n_triangles"] == 0 and sc.info()["n_spheres"] == 0


def test_camera_matches_reference_outputs(esc, golden_dir):
    with np.load(golden_dir + "/ref_pieces.npz") as f:
        z = {k: f[k] for k in f.files}
    for i in range(len(z["cam_vfov"])):
        cam = esc.Camera(z["cam_eye"][i], z["cam_look"][i], (0, 1, 0), z["cam_vfov"][i],
                         z["cam_aspect"][i])
        v = cam.vectors()
        got = np.concatenate([v["origin"], v["lower_left_corner"], v["horizontal"], v["vertical"]])
        assert np.array_equal(got.view(np.uint32), z["cam_vectors"][i].view(np.uint32)), i


def test_synthetic_scenes_are_frozen(esc):
    """BASELINE configs: counts and a few exact values, so the workload cannot drift"""
    for cfg, n in (("c2", 100), ("c3", 1000), ("c4", 10000)):
        sc = esc.Scene.synthetic(cfg)
        assert sc.info() == {"n_geometry": 2, "n_lights": 1, "n_triangles": 3, "n_spheres": n}
    s, m = esc.Scene.synthetic("c4").spheres()
    assert np.allclose(s[0], [-7.932042, 2.0471272, -14.4721775, 0.11074812], rtol=0, atol=1e-6)
    assert s[:, 3].min() >= 0.05 and s[:, 3].max() <= 0.2 and (m[:, 6:12] == 0).all()
    c5 = esc.Scene.synthetic("c5")
    assert c5.info() == {"n_geometry": 2, "n_lights": 1, "n_triangles": 100353, "n_spheres": 0}
    eye, look = esc.synthetic_view()
    assert eye.tolist() == [0, 3, 6] and look.tolist() == [0, 2, -8]


# ------------------------------------------------------------------ flatten (flatten_iscp.cpp)
def test_flatten_ispc(esc):
    sc = ol.scene_to_product(ol.load_dump("CornellBox-Original"))
    flat = sc.flatten_ispc(False)
    assert (flat.num_triangles, flat.num_lights, flat.num_light_triangles) == (36, 1, 2)
    d = ol.load_dump("CornellBox-Original")
    k = 0
    for gi, g in enumerate(d["geometry"]):
        for f, face in enumerate(g["face_index"]):
            t = flat.triangles[k]
            assert (t.geom_id, t.prim_id, t.has_normals, t.is_light) == (gi, f, 0, int(gi == 6))
            assert np.array_equal(np.array(t.vertices, np.float32), g["vertex"][face])
            assert np.array(t.kd, np.float32).tolist() == g["material"][3:6].tolist()
            k += 1
    L = flat.lights[0]
    assert L.geom_id == 6 and [L.light_faces[i] for i in range(L.num_light_faces)] == [0, 1]
    srt = sc.flatten_ispc(True)  # flatten_iscp.cpp:110: ascending centroid x
    cx = [sum(srt.triangles[i].vertices[v][0] for v in range(3)) for i in range(36)]
    assert all(np.float32(a / 3) <= np.float32(b / 3) for a, b in zip(cx, cx[1:]))
    with pytest.raises(esc.EscError):
        esc.Scene.synthetic("c2").flatten_ispc()  # spheres have no ispc_triangle form


def test_check_flat_rejects_what_the_kernels_could_not_survive(esc):
    """esc_check_flat (the staging half of `trace`, no GPU): a light without faces would reach
    `hash % n_faces` on the device; out-of-range face indices and negative geom ids
    would index outside the tables."""
    flat = ol.scene_to_product(ol.load_dump("CornellBox-Original")).flatten_ispc(False)
    flat.check()
    L = flat.lights[0]
    keep = (L.num_light_faces, L.light_faces[0], flat.triangles[3].geom_id)
    for bad in (0, -1):
        L.num_light_faces = bad
        with pytest.raises(esc.EscError, match="num_light_faces"):
            flat.check()
    L.num_light_faces = keep[0]
    L.light_faces[0] = flat.num_light_triangles  # one past the end
    with pytest.raises(esc.EscError, match="out of range"):
        flat.check()
    L.light_faces[0] = keep[1]
    flat.triangles[3].geom_id = -2
    with pytest.raises(esc.EscError, match="geom_id"):
        flat.check()
    flat.triangles[3].geom_id = keep[2]
    flat.check()
    # the scene path cannot even hold a light without faces: add_geometry refuses it
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    with pytest.raises(esc.EscError):
        esc.Scene().add_geometry(v, np.zeros((0, 3), np.uint32), ol.material13(ke=(5, 5, 5)))


# ------------------------------------------------------------------ PPM (main.cpp:658-689)
def test_ppm_writer_equals_reference_format(esc, tmp_path):
    rng = np.random.default_rng(3)
    img = rng.uniform(-0.2, 1.4, (37, 53, 3)).astype(np.float32)
    img[0, 0] = (1.0, 0.999999, 0.0)
    a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
    esc.write_ppm(a, np.maximum(img, 0))
    assert ol.oracle().orc_write_ppm(str(b).encode(), ol.fp(np.maximum(img, 0)), 53, 37) == 0
    assert a.read_bytes() == b.read_bytes()
    lines = a.read_text().split("\n")
    assert lines[:3] == ["P3", "53 37", "255"] and len(lines) == 3 + 37 * 53 + 1
    q = esc.quantise(np.maximum(img, 0))
    assert np.array_equal(q, ol.oracle_quantise(np.maximum(img, 0)))
    assert q[0, 0].tolist() == [255, 254, 0]  # clamp only > 1, truncation not rounding
    # first text row is the TOP row h = H-1
    assert lines[3] == " ".join(str(v) for v in q[36, 0])
    esc.write_ppm(tmp_path / "c.ppm", q)
    assert (tmp_path / "c.ppm").read_bytes() == a.read_bytes()


def test_viewer_cli_surface():
    """flags of main.cpp:430-535 are accepted; an unknown flag is rejected like :531-534"""
    exe = os.path.join(ROOT, "bin", "ESCViewer2021")
    if not os.path.exists(exe):
        pytest.skip("viewer not built")
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid Argument: --bogus" in r.stderr
    r = subprocess.run([exe, "-v", "1,2"], capture_output=True, text=True)
    assert r.returncode != 0 and "Error parsing view" in r.stderr


def test_host_code_under_sanitizers(tmp_path, golden_dir):
    """scene model, loader, synthetic scenes, flatten, PPM writer and BVH builder under ASan + UBSan (CPU
    build of the host half only; the GPU pool offers no sanitizer runs)"""
    exe = tmp_path / "host_sanitize"
    src = [os.path.join(ROOT, "tools", "host_sanitize.cpp")] + [
        os.path.join(ROOT, "esctp1raytracer_amd", "host", f)
        for f in ("host_core.cpp", "obj_loader.cpp", "synth.cpp", "accel_build.cpp")]
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                        "-fno-omit-frame-pointer", "-ffp-contract=off",
                        "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "esctp1raytracer_amd", "host"),
                        "-I" + os.path.join(ROOT, "esctp1raytracer_amd", "csrc")] + src +
                       ["-o", str(exe)], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not available")
    assert b.returncode == 0, b.stderr[-2000:]
    objs = [os.path.join(golden_dir, "scenes", n) for n in ("one.obj", "two.obj")]
    if os.path.isdir("/root/reference/src/models/cornell"):
        objs += sorted(os.path.join("/root/reference/src/models/cornell", f)
                       for f in os.listdir("/root/reference/src/models/cornell") if f.endswith(".obj"))
    r = subprocess.run([str(exe)] + objs, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "failures=0" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr
    assert "runtime error" not in r.stderr


def test_strip_partition_properties(esc):
    """the C ABI's strip arithmetic against the Python mirror, for arbitrary heights and rank
    counts: every row belongs to exactly one rank, rank 0 never holds fewer rows than another"""
    from hypothesis import given, settings, strategies as st_

    from esctp1raytracer_amd import multigpu

    @settings(max_examples=200, deadline=None)
    @given(H=st_.integers(1, 9000), world=st_.integers(1, 64), k=st_.integers(1, 8))
    def check(H, world, k):
        S = 8 * k
        rows = [esc.strip_local_rows(H, S, r, world) for r in range(world)]
        assert rows == [multigpu.local_rows(H, r, world, S) for r in range(world)]
        assert sum(rows) == H and max(rows) == rows[0] == multigpu.max_local_rows(H, world, S)
        seen = sorted(k2 for r in range(world) for k2 in multigpu.strips_of_rank(H, r, world, S))
        assert seen == list(range(multigpu.n_strips(H, S)))

    check()
    with pytest.raises(esc.EscError):
        esc.strip_local_rows(100, 12, 0, 2)  # strip height must be a multiple of 8


# ---------------------------------------------------------------- primitive groups, host side
@pytest.mark.parametrize("n", [1, 7, 8, 9, 63, 64, 65, 511, 512, 513, 1000, 4097, 20000])
def test_group_order_is_a_permutation_with_aligned_subtrees(n):
    """esc_group_order (csrc/rt_device.h SphGroups / TriGroups): the k-d order the groups are cut
    from.  A permutation; deterministic; and its runs of 8 / 64 / 512 are spatially tighter than
    runs of the identity order (that is the whole point of sorting)."""
    import numpy as np
    from esctp1raytracer_amd import _capi
    lib = _capi.load()
    rng = np.random.default_rng(n)
    xyz = np.ascontiguousarray(rng.uniform(-10, 10, (n, 3)), np.float32)
    order = np.zeros(n, np.int32)
    FP, IP = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    assert lib.esc_group_order(xyz.ctypes.data_as(FP), n, 8, 64, 512, order.ctypes.data_as(IP)) == 0
    assert sorted(order.tolist()) == list(range(n))
    again = np.zeros(n, np.int32)
    assert lib.esc_group_order(xyz.ctypes.data_as(FP), n, 8, 64, 512, again.ctypes.data_as(IP)) == 0
    assert (order == again).all()
    if n >= 512:
        def spread(idx, run):
            m = (len(idx) // run) * run
            p = xyz[idx[:m]].reshape(-1, run, 3)
            return float((p.max(axis=1) - p.min(axis=1)).max(axis=1).mean())
        ident = np.arange(n)
        for run in (8, 64, 512):
            if 4 * run <= n:
                assert spread(order, run) < 0.8 * spread(ident, run)
    # bad arguments are refused
    assert lib.esc_group_order(xyz.ctypes.data_as(FP), n, 8, 60, 512, order.ctypes.data_as(IP)) < 0
    assert lib.esc_group_order(xyz.ctypes.data_as(FP), 0, 8, 64, 512, order.ctypes.data_as(IP)) < 0


def test_sphere_group_record_holds_every_member():
    """esc_sphere_group_record: rgeo >= r_i + |c_i - C| for every member, computed against the
    fp32 centre that is stored, and not wastefully larger than the members' own extent."""
    import numpy as np
    from esctp1raytracer_amd import _capi
    lib = _capi.load()
    FP = C.POINTER(C.c_float)
    rng = np.random.default_rng(5)
    for case in range(300):
        cnt = int(rng.integers(1, 65))
        scale = 10.0 ** rng.uniform(-2, 3)
        off = rng.uniform(-1, 1, 3) * scale * rng.choice([0, 1, 100])
        c = rng.normal(0, 1, (cnt, 3)) * scale + off
        r = 10.0 ** rng.uniform(-3, 0, cnt) * scale
        s = np.ascontiguousarray(np.concatenate([c, (r * r)[:, None]], 1), np.float32)
        rec = np.zeros(4, np.float32)
        assert lib.esc_sphere_group_record(s.ctypes.data_as(FP), cnt, rec.ctypes.data_as(FP)) == 0
        C64 = rec[:3].astype(np.float64)
        reach = np.sqrt(s[:, 3].astype(np.float64)) + np.linalg.norm(s[:, :3].astype(np.float64) - C64, axis=1)
        assert (reach <= float(rec[3])).all(), case
        ext = (s[:, :3].astype(np.float64) + np.sqrt(s[:, 3].astype(np.float64))[:, None]).max(0) - \
              (s[:, :3].astype(np.float64) - np.sqrt(s[:, 3].astype(np.float64))[:, None]).min(0)
        assert float(rec[3]) <= 0.87 * np.linalg.norm(ext) * (1 + 1e-5) + 1e-30  # <= half the diagonal


def test_triangle_group_record_bounds():
    """esc_tri_group_record: rgeo >= rho_t + |v - C| over the members' vertices, smax >= the sine
    between the stored axis and every member's normal line, rext >= |v0 - C|_1 + |e1|_1 + |e2|_1,
    b0 / b1 > 0; a sliver or a zero-area member makes the group `always` open."""
    import numpy as np
    from esctp1raytracer_amd import _capi
    lib = _capi.load()
    FP = C.POINTER(C.c_float)
    rng = np.random.default_rng(9)
    for case in range(300):
        cnt = int(rng.integers(1, 129))
        scale = 10.0 ** rng.uniform(-2, 2)
        base = rng.normal(0, 1, 3)
        v0 = rng.normal(0, 1, (cnt, 3)) * scale
        e1 = (rng.normal(0, 1, (cnt, 3)) * 0.2 + np.cross(base, [1, 0.3, 0.2])) * scale * 0.1
        e2 = (rng.normal(0, 1, (cnt, 3)) * 0.2 + np.cross(base, [0.1, 1, 0.4])) * scale * 0.1
        t = np.ascontiguousarray(np.concatenate([v0, e1, e2], 1), np.float32)
        rec = np.zeros(12, np.float32)
        assert lib.esc_tri_group_record(t.ctypes.data_as(FP), cnt, rec.ctypes.data_as(FP)) == 0
        assert rec[11] == 0
        t64 = t.astype(np.float64)
        V0, E1, E2 = t64[:, 0:3], t64[:, 3:6], t64[:, 6:9]
        Cc = rec[:3].astype(np.float64)
        G = V0 + (E1 + E2) / 3
        rho = np.sqrt(np.maximum(((G - V0) ** 2).sum(1), np.maximum(((G - V0 - E1) ** 2).sum(1),
                                                                   ((G - V0 - E2) ** 2).sum(1))))
        for V in (V0, V0 + E1, V0 + E2):
            assert (np.linalg.norm(V - Cc, axis=1) + rho <= float(rec[3])).all(), case
        nrm = np.cross(E2, E1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        a = rec[4:7].astype(np.float64)
        sines = np.linalg.norm(np.cross(a / np.linalg.norm(a), nrm), axis=1)
        assert (sines <= float(rec[7])).all(), case
        ext = np.abs(V0 - Cc).sum(1) + np.abs(E1).sum(1) + np.abs(E2).sum(1)
        assert (ext <= float(rec[8])).all(), case
        assert rec[9] > 0 and rec[10] > 0
    # a collinear member: always open
    t = np.ascontiguousarray([[0, 0, 0, 1, 0, 0, 0, 1, 0], [5, 5, 5, 1, 1, 1, 2, 2, 2]], np.float32)
    rec = np.zeros(12, np.float32)
    assert lib.esc_tri_group_record(t.ctypes.data_as(FP), 2, rec.ctypes.data_as(FP)) == 0
    assert rec[11] != 0


def test_committed_profile_was_measured_on_these_sources():
    """profiles/current.json feeds bench.py's roofline.traffic and `valu` view; bench.py drops
    it (traffic: null) when its stamp -- sha256 over csrc/* and the Makefile -- is not the tree's.
    The committed state must not be in that condition."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    ns = {"os": os, "ROOT": root}
    exec(src[src.index("def source_stamp():"):src.index("def committed_profile")], ns)
    cur = json.load(open(os.path.join(root, "profiles", "current.json")))
    assert cur["source_stamp"] == ns["source_stamp"](), \
        "csrc/ or the Makefile changed after the last tools/profile.sh + summarize_prof.py --current run"
    # both sweeps are profiled: the default (culled) path and the linear brute-force path
    assert set(cur["paths"]) == {"culled", "linear"}
    for path in cur["paths"].values():
        assert path["hbm_bytes_per_frame"] > 0
        for kern in path["kernels"].values():
            for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU",
                      "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
                assert kern.get(c, 0) > 0, c
