"""HIP path vs the CPU oracle, through the C ABI, on a real MI355X.

Bar (BASELINE.json north_star): pixel-for-pixel after 8-bit PPM quantisation, fp colour
within 1e-5 before it.  Everything here is checked BIT-EXACT on the fp32 framebuffer except
scenes with ks != 0, where device powf and glibc powf may differ in the last ulp and the
1e-5 tolerance of north_star applies (written at that test).
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def esc():
    import esctp1raytracer_amd as m
    return m


@pytest.fixture(scope="module")
def renderer(esc):
    r = esc.Renderer(0)
    yield r
    r.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(gpu, ref, what):
    nb = int((bits(gpu) != bits(ref)).sum())
    assert nb == 0, f"{what}: {nb} of {ref.size} fp32 values differ, max abs " \
                    f"{float(np.abs(gpu - ref).max())}"


def render_both(esc, renderer, d, eye, look, W, H, **kw):
    sc = ol.scene_to_product(d)
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    okw = {k: v for k, v in kw.items() if k in ("shadows", "face_mode", "fixed_face", "seed")}
    # the product's ESC_FACE_* / the checker's ORC_FACE_* share values 0 / 1
    ref = ol.oracle_render(d, eye, look, W, H, threads=8, **okw)
    gpu, u8 = renderer.render(cam, W, H, want_u8=True, **kw)
    return gpu, u8, ref


# ---------------------------------------------------------------- reference-pinned scenes
@pytest.mark.parametrize("px", [1, 2, 4])
@pytest.mark.parametrize("stage", ["smem", "lds"])
@pytest.mark.parametrize("name,eye", [("one", (0, 1, 3)), ("two", (0, 1, 3)),
                                      ("CornellBox-Original", (0, 1, 2)),
                                      ("CornellBox-Empty-CO", (0, 1, 3)),
                                      ("cornell_box", (0, 1, 3))])
def test_triangle_scenes_bit_exact(esc, renderer, name, eye, stage, px):
    d = ol.load_dump(name)
    st = esc.ESC_STAGE_SMEM if stage == "smem" else esc.ESC_STAGE_LDS
    gpu, u8, ref = render_both(esc, renderer, d, eye, (0, 1, 0), 160, 90, stage=st, px=px)
    assert_bit_equal(gpu, ref, f"{name}/{stage}/px{px}")
    assert np.array_equal(u8, ol.oracle_quantise(ref))
    assert ref.sum() > 0


def test_committed_golden_frames(esc, renderer, golden_dir):
    z = np.load(golden_dir + "/frames.npz")
    for name, eye, face in (("one", (0, 1, 3), 0), ("two", (0, 1, 3), 0),
                            ("CornellBox-Original", (0, 1, 2), 0),
                            ("CornellBox-Original_face1", (0, 1, 2), 1)):
        d = ol.load_dump(name.replace("_face1", ""))
        renderer.upload(ol.scene_to_product(d))
        cam = esc.Camera.for_image(eye, (0, 1, 0), 96, 72)
        gpu = renderer.render(cam, 96, 72, fixed_face=face)
        assert_bit_equal(gpu, z[name], name)


def test_one_1024x768_ppm_md5(esc, renderer, tmp_path):
    """The reference viewer's own PPM for scene `one`."""
    import hashlib
    d = ol.load_dump("one")
    renderer.upload(ol.scene_to_product(d))
    cam = esc.Camera.for_image((0, 1, 3), (0, 1, 0), 1024, 768)
    img, u8 = renderer.render(cam, 1024, 768, want_u8=True)
    p = tmp_path / "one.ppm"
    esc.write_ppm(p, img)
    assert hashlib.md5(p.read_bytes()).hexdigest() == "b10e1cb14f839129bd111670002cfb0b"
    p2 = tmp_path / "one_u8.ppm"
    esc.write_ppm(p2, u8)
    assert p2.read_bytes() == p.read_bytes()


def test_two_1024x768_ppm_md5(esc, renderer, tmp_path):
    """The reference viewer's PPM for scene `two` (two lights + smooth normals: quirks S1 + S3,
    main.cpp:307,310,737,757-772); MD5 measured from the unmodified reference."""
    import hashlib
    d = ol.load_dump("two")
    renderer.upload(ol.scene_to_product(d))
    cam = esc.Camera.for_image((0, 1, 3), (0, 1, 0), 1024, 768)
    for stage in (esc.ESC_STAGE_AUTO, esc.ESC_STAGE_LDS, esc.ESC_STAGE_BVH):
        img, u8 = renderer.render(cam, 1024, 768, want_u8=True, stage=stage)
        p = tmp_path / f"two_{stage}.ppm"
        esc.write_ppm(p, img)
        assert hashlib.md5(p.read_bytes()).hexdigest() == "c8137a8d70d8de0ad6a001d41be4e0e1"
        p2 = tmp_path / f"two_u8_{stage}.ppm"
        esc.write_ppm(p2, u8)
        assert p2.read_bytes() == p.read_bytes()


def test_config1_default_scene_full_size(esc, renderer):
    """BASELINE.json config 1: CornellBox-Original, eye 0,1,2, look 0,1,0, 1024x768
    (scripts/run.sh:28-30; geometry = what the reference's loader returns)."""
    d = ol.load_dump("CornellBox-Original")
    gpu, u8, ref = render_both(esc, renderer, d, (0, 1, 2), (0, 1, 0), 1024, 768,
                               face_mode=esc.ESC_FACE_HASH, seed=0)
    assert_bit_equal(gpu, ref, "config 1")
    assert np.array_equal(u8, ol.oracle_quantise(ref))
    assert int((ref.sum(axis=2) > 0).sum()) > 300000


def test_viewer_binary_modes(esc, tmp_path, golden_dir):
    """bin/ESCViewer2021 through every mode flag of scripts/run.sh: identical PPM, equal to the
    library's frame (the --ispc mode goes through flatten + the `trace` symbol)."""
    import os
    import subprocess
    exe = os.path.join(ol.ROOT, "bin", "ESCViewer2021")
    if not os.path.exists(exe):
        pytest.skip("viewer not built")
    obj = os.path.join(golden_dir, "scenes", "one.obj")
    outs = []
    for i, flags in enumerate(([], ["--thread"], ["--bvh"], ["--ispc"], ["--gpus", "3"])):
        out = tmp_path / f"o{i}.ppm"
        r = subprocess.run([exe, "-m", obj, "-v", "0,1,3", "-l", "0,1,0", "-o", str(out)] + flags,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Duration" in r.stderr and "Rendered image in:" in r.stdout
        outs.append(out.read_bytes())
    assert all(o == outs[0] for o in outs)
    import hashlib
    assert hashlib.md5(outs[0]).hexdigest() == "b10e1cb14f839129bd111670002cfb0b"
    # scene `two` through the loader + viewer: two lights, per-vertex normals (S1 + S3)
    obj2 = os.path.join(golden_dir, "scenes", "two.obj")
    for i, flags in enumerate(([], ["--bvh"], ["--ispc"])):
        out = tmp_path / f"t{i}.ppm"
        r = subprocess.run([exe, "-m", obj2, "-v", "0,1,3", "-l", "0,1,0", "-o", str(out)] + flags,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert hashlib.md5(out.read_bytes()).hexdigest() == "c8137a8d70d8de0ad6a001d41be4e0e1"


def test_multi_face_light_hash_and_membership(esc, renderer):
    """2-face light (every bundled Cornell scene): hashed face choice matches the oracle bit
    for bit, and each pixel equals the face-0 or the face-1 render."""
    d = ol.load_dump("CornellBox-Original")
    gpu, _, ref = render_both(esc, renderer, d, (0, 1, 2), (0, 1, 0), 128, 96,
                              face_mode=esc.ESC_FACE_HASH, seed=12345)
    assert_bit_equal(gpu, ref, "hash face")
    f0 = ol.oracle_render(d, (0, 1, 2), (0, 1, 0), 128, 96, fixed_face=0)
    f1 = ol.oracle_render(d, (0, 1, 2), (0, 1, 0), 128, 96, fixed_face=1)
    px = lambda a: bits(a).reshape(-1, 3)
    is0 = (px(gpu) == px(f0)).all(axis=1)
    is1 = (px(gpu) == px(f1)).all(axis=1)
    assert (is0 | is1).all()
    assert (~is0).any() and (~is1).any()


def test_big_mesh_smooth_normals(esc, renderer):
    """CornellBox-Sphere: 2,188 triangles, per-vertex normals (quirk S1), ks != 0 on some
    materials -> powf differs device vs glibc by <= 1 ulp: tolerance 1e-5 relative to the
    pixel value, as north_star states; pixels of ks == 0 materials must still be bit exact."""
    d = ol.load_dump("CornellBox-Sphere")
    gpu, u8, ref = render_both(esc, renderer, d, (0, 1, 3), (0, 1, 0), 96, 72,
                               face_mode=esc.ESC_FACE_HASH, seed=7)
    assert np.allclose(gpu, ref, rtol=1e-5, atol=1e-5)
    frac_exact = float((bits(gpu) == bits(ref)).mean())
    assert frac_exact > 0.98, frac_exact
    assert (u8.astype(int) - ol.oracle_quantise(ref).astype(int)).__abs__().max() <= 1


# ---------------------------------------------------------------- sphere extension
def synthetic_dict(esc, config, n):
    sc = esc.Scene.synthetic(config, n)
    return sc, ol.scene_from_product(sc)


@pytest.mark.parametrize("px", [1, 2, 4])
@pytest.mark.parametrize("stage", ["smem", "lds"])
@pytest.mark.parametrize("config,n,shadows", [("c2", 100, False), ("c3", 1000, True),
                                              ("c4", 613, True), ("c4", 10000, True)])
def test_sphere_scenes_bit_exact(esc, renderer, config, n, shadows, stage, px):
    sc, d = synthetic_dict(esc, config, n)
    eye, look = esc.synthetic_view()
    W, H = (192, 108) if n < 5000 else (96, 54)
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    st = esc.ESC_STAGE_SMEM if stage == "smem" else esc.ESC_STAGE_LDS
    ref, rc = ol.oracle_render(d, eye, look, W, H, shadows=shadows, threads=8,
                               return_counters=True)
    # index order (the reference's): every counter equals the oracle's; default order (long
    # sphere lists are swept by decreasing solid angle for the LAST light): same image, the other
    # counters equal, and fewer any-hit tests
    for flags in (esc.ESC_RENDER_INDEX_ORDER, 0):
        renderer.reset_counters()
        gpu, u8 = renderer.render(cam, W, H, want_u8=True, shadows=shadows, stage=st, px=px,
                                  flags=flags)
        cnt = renderer.counters()
        assert_bit_equal(gpu, ref, f"{config}/{n}/{stage}/px{px}/flags{flags}")
        assert np.array_equal(u8, ol.oracle_quantise(ref))
        lane_tests = cnt.pop("anyhit_lane_tests")
        if flags == esc.ESC_RENDER_INDEX_ORDER:
            assert cnt == rc
            if shadows and stage == "smem":
                assert lane_tests >= rc["anyhit_tests"]  # lanes spent >= tests the reference needs
        else:
            tests = cnt.pop("anyhit_tests")
            assert cnt == {k: v for k, v in rc.items() if k != "anyhit_tests"}
            assert tests <= rc["anyhit_tests"]
    assert 0 < rc["hit_pixels"] < W * H


@pytest.mark.parametrize("config,W,H", [("c3", 3840, 2160), ("c4", 3840, 2160)])
def test_filter_equals_exact_only_full_size(esc, renderer, config, W, H):
    """The brute-force kernels run a conservative FMA filter per (ray, sphere) and the reference
    arithmetic only where the filter cannot rule a hit out (csrc/rt_brute.h "FILTERS").  Whole
    BASELINE-size frames: filtered == ESC_RENDER_EXACT_ONLY (the reference arithmetic for every
    pair) in every fp32 value, every quantised byte and every counter -- except anyhit_tests,
    which counts what each sweep order executed (index order, solid angle, sphere groups) and is
    compared with the oracle's in index order elsewhere."""
    import torch
    sc = esc.Scene.synthetic(config)
    eye, look = esc.synthetic_view()
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    out = []
    for flags in (0, esc.ESC_RENDER_EXACT_ONLY, esc.ESC_RENDER_INDEX_ORDER):
        f32 = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
        u8 = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda:0")
        renderer.reset_counters()
        renderer.render_rows(cam, W, H, 0, H, out_f32=f32, out_u8=u8, flags=flags)
        cnt = renderer.counters()
        cnt.pop("anyhit_lane_tests")
        cnt.pop("anyhit_tests")
        out.append((f32, u8, cnt))
    for other in out[1:]:
        assert int((out[0][0].view(torch.int32) != other[0].view(torch.int32)).sum().item()) == 0
        assert bool((out[0][1] == other[1]).all().item())
        assert out[0][2] == other[2]
    assert float(out[0][0].sum().item()) > 0


@pytest.mark.parametrize("case", ["far from the world origin", "tiny and huge radii",
                                  "camera inside a sphere", "sphere centred on the camera",
                                  "rays grazing silhouettes", "1e4 scale"])
def test_filter_adversarial_scenes_vs_oracle(esc, renderer, case):
    """Scenes chosen against the filters' margins: coordinates much larger than the radii (the
    margins scale with squared distances), radii spanning 1e-3..1e2, origins inside spheres
    (cc < 0), grazing rays.  Filtered frame == oracle, bit for bit, with 1 and 2 lights."""
    rng = np.random.default_rng(len(case) * 7 + ord(case[0]))
    n = 160
    off = np.zeros(3)
    scale = 1.0
    eye, look = np.array([0.0, 1.0, 6.0]), np.array([0.0, 1.0, 0.0])
    c = np.concatenate([rng.uniform(-3, 3, (n, 1)), rng.uniform(0.2, 3.5, (n, 1)),
                        rng.uniform(-4, 2, (n, 1))], axis=1)
    r = rng.uniform(0.05, 0.4, n)
    if case == "far from the world origin":
        off = np.array([700.0, -300.0, 1500.0])
    elif case == "tiny and huge radii":
        r = 10.0 ** rng.uniform(-3, 0, n)
        c[0], r[0] = (0.0, -100.0, -3.0), 100.0  # a planet under everything
    elif case == "camera inside a sphere":
        c[0], r[0] = eye + (0.1, 0.05, -0.2), 1.5
        c[1], r[1] = eye, 40.0                     # and everything inside a big shell
    elif case == "sphere centred on the camera":
        c[0], r[0] = eye, 0.25
    elif case == "rays grazing silhouettes":
        # rows of equal spheres touching each other: every gap is a tangent configuration
        k = np.arange(n)
        c = np.stack([(k % 20) * 0.3 - 2.85, (k // 20) * 0.3 + 0.3, np.full(n, -1.0)], axis=1)
        r = np.full(n, 0.15)
    elif case == "1e4 scale":
        scale = 1.0e4
    c = (c + off) * scale
    r = r * scale
    eye, look = (eye + off) * scale, (look + off) * scale
    fl = (np.array([[-6, 0, 4], [6, 0, 4], [6, 0, -8], [-6, 0, -8]], float) + off) * scale
    l1 = (np.array([[-0.3, 7, -1], [0.3, 7, -1], [0, 7, -1.6]], float) + off) * scale
    l2 = (np.array([[4, 5, 2], [4.4, 5, 2], [4, 5.4, 2.2]], float) + off) * scale
    for lights in ([l1], [l1, l2]):
        geoms = [{"vertex": fl[[0, 1, 2, 0, 2, 3]].astype(np.float32),
                  "face_index": np.arange(6).reshape(2, 3), "material": ol.WHITE}]
        for lt in lights:
            geoms.append({"vertex": lt.astype(np.float32), "face_index": np.array([[0, 1, 2]]),
                          "material": ol.LIGHT_A})
        sph = np.concatenate([c, r[:, None]], axis=1).astype(np.float32)
        mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.2, 0.9, (n, 3))])
        d = ol.scene_dict(geoms, sph, mats)
        W, H = 224, 128
        for px in (1, 2):
            gpu, u8, ref = render_both(esc, renderer, d, tuple(eye), tuple(look), W, H, px=px)
            assert_bit_equal(gpu, ref, f"filter/{case}/{len(lights)} lights/px{px}")
        exact = renderer.render(esc.Camera.for_image(tuple(eye), tuple(look), W, H), W, H,
                                flags=esc.ESC_RENDER_EXACT_ONLY)
        assert_bit_equal(exact, ref, f"exact-only/{case}/{len(lights)} lights")
        assert ref.sum() > 0


@pytest.mark.parametrize("case", ["equal t: duplicated spheres", "tight clusters far apart",
                                  "camera between the members of a group", "far specks",
                                  "one group and a bit", "collinear centres"])
def test_sphere_groups_vs_oracle(esc, renderer, case):
    """From 64 spheres up the brute-force kernels test bounding spheres of k-d ordered runs of 8
    spheres first (csrc/rt_device.h SphGroups).  Scenes against what that adds: hits at EQUAL t on
    spheres with different indices and materials (the reference keeps the lower index; the groups
    are not swept in index order), groups whose bounding sphere holds the camera, specks whose
    accepts are rounding noise, partly filled last groups, degenerate splits.  Frame == oracle bit
    for bit with 1, 2 and 3 lights, 1 and 2 pixels per lane; == index order == exact-only."""
    rng = np.random.default_rng(sum(map(ord, case)))
    eye, look = np.array([0.0, 1.0, 6.0]), np.array([0.0, 1.0, 0.0])
    n = 200
    c = np.concatenate([rng.uniform(-3, 3, (n, 1)), rng.uniform(0.2, 3.5, (n, 1)),
                        rng.uniform(-4, 2, (n, 1))], axis=1)
    r = rng.uniform(0.05, 0.4, n)
    if case == "equal t: duplicated spheres":
        c, r = np.concatenate([c[:70]] * 3), np.concatenate([r[:70]] * 3)  # 3 copies, 3 materials
        perm = rng.permutation(len(r))
        c, r = c[perm], r[perm]
    elif case == "tight clusters far apart":
        centres = rng.uniform(-1, 1, (25, 3)) * (40, 15, 40) + (0, 16, -45)
        c = (centres[:, None, :] + rng.normal(0, 0.3, (25, 8, 3))).reshape(-1, 3)
        r = rng.uniform(0.1, 0.5, len(c))
    elif case == "camera between the members of a group":
        v = rng.normal(0, 1, (16, 3))
        c[:16] = eye + v / np.linalg.norm(v, axis=1)[:, None] * 0.8  # a shell around the camera
        r[:16] = 0.2
    elif case == "far specks":
        c[:120] = rng.uniform(-1, 1, (120, 3)) * (300, 100, 50) + (0, 100, -900)
        r[:120] = 10.0 ** rng.uniform(-4, -1, 120)
    elif case == "one group and a bit":
        c, r = c[:67], r[:67]
    elif case == "collinear centres":
        k = np.arange(n)
        c = np.stack([k * 0.03 - 3.0, np.full(n, 1.0), np.full(n, -1.0)], axis=1)
        r = np.full(n, 0.1)
    n = len(r)
    fl = np.array([[-6, 0, 4], [6, 0, 4], [6, 0, -8], [-6, 0, -8]], float)
    l1 = np.array([[-0.3, 7, -1], [0.3, 7, -1], [0, 7, -1.6]], float)
    l2 = np.array([[4, 5, 2], [4.4, 5, 2], [4, 5.4, 2.2]], float)
    l3 = np.array([[-5, 3, 3], [-5, 3.3, 3], [-4.8, 3, 3.3]], float)
    sph = np.concatenate([c, r[:, None]], axis=1).astype(np.float32)
    mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.2, 0.9, (n, 3))])
    for lights in ([l1], [l1, l2], [l1, l2, l3]):
        geoms = [{"vertex": fl[[0, 1, 2, 0, 2, 3]].astype(np.float32),
                  "face_index": np.arange(6).reshape(2, 3), "material": ol.WHITE}]
        for lt in lights:
            geoms.append({"vertex": lt.astype(np.float32), "face_index": np.array([[0, 1, 2]]),
                          "material": ol.LIGHT_A})
        d = ol.scene_dict(geoms, sph, mats)
        W, H = 224, 128
        for px in (1, 2):
            gpu, u8, ref = render_both(esc, renderer, d, tuple(eye), tuple(look), W, H, px=px)
            assert_bit_equal(gpu, ref, f"groups/{case}/{len(lights)} lights/px{px}")
        cam = esc.Camera.for_image(tuple(eye), tuple(look), W, H)
        for flags in (esc.ESC_RENDER_INDEX_ORDER, esc.ESC_RENDER_EXACT_ONLY,
                      esc.ESC_RENDER_SHADE_QUEUE):
            other = renderer.render(cam, W, H, flags=flags)
            assert_bit_equal(other, ref, f"groups/{case}/{len(lights)} lights/flags{flags}")
        assert ref.sum() > 0


def _grid_mesh(nx, nz, x0, x1, z0, z1, height):
    """(nx x nz quads) x 2 triangles over [x0,x1] x [z0,z1], y = height(x, z), facing +y; float64
    vertices [n, 3, 3]"""
    xs, zs = np.linspace(x0, x1, nx + 1), np.linspace(z0, z1, nz + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    P = np.stack([X, height(X, Z), Z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    return np.concatenate([np.stack([a, c, b], axis=2).reshape(-1, 3, 3),   # normals up (+y)
                           np.stack([a, d, c], axis=2).reshape(-1, 3, 3)])


@pytest.mark.parametrize("case", ["camera in the plane of a tessellated floor",
                                  "light flush with a tessellated ceiling",
                                  "equal t: the same mesh twice", "slivers and collapsed triangles",
                                  "a small mesh far away", "bumps seen at a grazing angle"])
def test_triangle_groups_vs_oracle(esc, renderer, case):
    """From 64 triangles up the brute-force kernels test a bounding sphere and a normal cone per
    k-d ordered run of 8 / 64 triangles first (csrc/rt_device.h TriGroups).  Scenes against the
    cone's "nearly parallel" escape and the ordering: rays and origins IN the plane of many
    coplanar triangles (accepts there are rounding noise, and the groups must still let every
    one of them through), hits at equal t on triangles of different geometries, slivers and
    zero-area triangles inside groups, a far mesh of tiny triangles, silhouettes of a bumpy mesh.
    Frame == oracle bit for bit, 1-3 lights; == index order == exact-only == queue form."""
    rng = np.random.default_rng(sum(map(ord, case)))
    eye, look = np.array([0.0, 1.0, 6.0]), np.array([0.0, 1.0, 0.0])
    flat = lambda X, Z: np.zeros_like(X)
    meshes = [(_grid_mesh(12, 12, -6, 6, -8, 4, flat), ol.WHITE)]
    l1 = np.array([[-0.3, 7, -1], [0.3, 7, -1], [0, 7, -1.6]], float)
    if case == "camera in the plane of a tessellated floor":
        eye, look = np.array([0.3, 0.0, 3.7]), np.array([0.0, 0.0, -2.0])  # y == 0 exactly
        meshes.append((_grid_mesh(6, 6, -2, 2, -4, -1, lambda X, Z: 0.5 + 0.3 * np.sin(X) * np.cos(Z)),
                       ol.material13(ka=(0.3, 0.6, 0.8), kd=(0.3, 0.6, 0.8))))
    elif case == "light flush with a tessellated ceiling":
        meshes.append((_grid_mesh(10, 10, -6, 6, -8, 4, lambda X, Z: np.full_like(X, 7.0))[:, ::-1],
                       ol.material13(ka=(0.7, 0.7, 0.6), kd=(0.7, 0.7, 0.6))))
        eye, look = np.array([0.0, 0.5, 3.5]), np.array([0.0, 5.5, -2.0])  # looking up at it
    elif case == "equal t: the same mesh twice":
        bump = _grid_mesh(9, 9, -3, 3, -5, 1, lambda X, Z: 0.6 + 0.5 * np.sin(1.3 * X) * np.cos(0.9 * Z))
        meshes.append((bump, ol.material13(ka=(0.8, 0.2, 0.2), kd=(0.8, 0.2, 0.2))))
        meshes.append((bump[rng.permutation(len(bump))], ol.material13(ka=(0.2, 0.8, 0.2), kd=(0.2, 0.8, 0.2))))
    elif case == "slivers and collapsed triangles":
        m = _grid_mesh(10, 10, -3, 3, -5, 1, lambda X, Z: 0.8 + 0.4 * np.cos(X + Z))
        m[::7, 2] = m[::7, 0] + (m[::7, 1] - m[::7, 0]) * 0.5      # collinear: zero area
        m[3::11, 2] = m[3::11, 1]                                  # two equal vertices
        m[5::13, 2] = m[5::13, 0] + (m[5::13, 1] - m[5::13, 0]) * 0.3 + 1e-6  # slivers
        meshes.append((m, ol.material13(ka=(0.6, 0.5, 0.9), kd=(0.6, 0.5, 0.9))))
    elif case == "a small mesh far away":
        meshes.append((_grid_mesh(12, 12, -0.4, 0.4, -300.4, -299.6,
                                  lambda X, Z: 40 + 0.2 * np.sin(9 * X) * np.cos(7 * Z)),
                       ol.material13(ka=(0.9, 0.9, 0.2), kd=(0.9, 0.9, 0.2))))
        look = np.array([0.0, 40.0 * 6 / 306 + 1.0, 0.0])
    elif case == "bumps seen at a grazing angle":
        eye, look = np.array([0.0, 0.9, 5.0]), np.array([0.0, 0.3, -3.0])
        meshes.append((_grid_mesh(24, 24, -5, 5, -7, 3, lambda X, Z: 0.4 * np.sin(1.7 * X) * np.cos(1.3 * Z) + 0.41),
                       ol.material13(ka=(0.4, 0.7, 0.5), kd=(0.4, 0.7, 0.5))))
    l2 = np.array([[4, 5, 2], [4.4, 5, 2], [4, 5.4, 2.2]], float)
    l3 = np.array([[-5, 3, 3], [-5, 3.3, 3], [-4.8, 3, 3.3]], float)
    for lights in ([l1], [l1, l2], [l1, l2, l3]):
        geoms = [{"vertex": m.reshape(-1, 3).astype(np.float32),
                  "face_index": np.arange(3 * len(m)).reshape(-1, 3), "material": mat}
                 for m, mat in meshes]
        for lt in lights:
            geoms.append({"vertex": lt.astype(np.float32), "face_index": np.array([[0, 1, 2]]),
                          "material": ol.LIGHT_A})
        d = ol.scene_dict(geoms, np.zeros((0, 4), np.float32), np.zeros((0, 13), np.float32))
        W, H = 224, 128
        for px in (1, 2):
            gpu, u8, ref = render_both(esc, renderer, d, tuple(eye), tuple(look), W, H, px=px)
            assert_bit_equal(gpu, ref, f"tri groups/{case}/{len(lights)} lights/px{px}")
        cam = esc.Camera.for_image(tuple(eye), tuple(look), W, H)
        for flags in (esc.ESC_RENDER_INDEX_ORDER, esc.ESC_RENDER_EXACT_ONLY,
                      esc.ESC_RENDER_SHADE_QUEUE):
            other = renderer.render(cam, W, H, flags=flags)
            assert_bit_equal(other, ref, f"tri groups/{case}/{len(lights)} lights/flags{flags}")
        assert ref.sum() > 0


def test_groups_on_large_tables_vs_oracle(esc, renderer):
    """Many hyper-groups of both kinds in one scene -- 150,000 spheres and a 120,000-triangle
    mesh (294 + 118 hyper-groups, partly filled last ones, pad records up to whole sweep steps),
    two lights (the first swept in first-occluder mode) -- on a small frame: == oracle bit for
    bit, == index order."""
    rng = np.random.default_rng(2024)
    n = 150_000
    sph = np.concatenate([rng.uniform(-8, 8, (n, 1)), rng.uniform(0.3, 6, (n, 1)),
                          rng.uniform(-20, -2, (n, 1)), rng.uniform(0.01, 0.05, (n, 1))], 1)
    mats = np.tile(ol.material13(ka=(0.5, 0.6, 0.7), kd=(0.5, 0.6, 0.7)), (n, 1))
    mesh = _grid_mesh(300, 200, -12, 12, -24, 4, lambda X, Z: 0.25 * np.sin(1.1 * X) * np.cos(0.8 * Z))
    l1 = np.array([[-0.5, 12, -9.5], [0, 12, -10.5], [0.5, 12, -9.5]], float)
    l2 = np.array([[6, 9, -2], [6.5, 9, -2], [6, 9.5, -2.3]], float)
    geoms = [{"vertex": mesh.reshape(-1, 3).astype(np.float32),
              "face_index": np.arange(3 * len(mesh)).reshape(-1, 3), "material": ol.WHITE}]
    for lt in (l1, l2):
        geoms.append({"vertex": lt.astype(np.float32), "face_index": np.array([[0, 1, 2]]),
                      "material": ol.LIGHT_A})
    d = ol.scene_dict(geoms, sph.astype(np.float32), mats)
    eye, look = esc.synthetic_view()
    W, H = 96, 54
    gpu, u8, ref = render_both(esc, renderer, d, eye, look, W, H)
    assert_bit_equal(gpu, ref, "large tables")
    assert np.array_equal(u8, ol.oracle_quantise(ref))
    cam = esc.Camera.for_image(eye, look, W, H)
    other = renderer.render(cam, W, H, flags=esc.ESC_RENDER_INDEX_ORDER)
    assert_bit_equal(other, ref, "large tables / index order")
    assert ref.sum() > 0


@pytest.mark.parametrize("form", ["queue", "fused"])
def test_both_shading_forms_on_small_frames(esc, renderer, form):
    """The queue form is chosen by default only for long primitive lists on large bands; force
    each form (ESC_RENDER_SHADE_QUEUE / _FUSED) through scenes that exercise what differs between
    them: two and three lights (t carried from light to light, quirk S3), triangles + spheres,
    normals, a multi-face light with the hashed face choice, ragged sizes, cyclic strips."""
    import torch
    from esctp1raytracer_amd import multigpu
    flags = esc.ESC_RENDER_SHADE_QUEUE if form == "queue" else esc.ESC_RENDER_SHADE_FUSED
    rng = np.random.default_rng(17)
    base = ol.load_dump("two")  # floor, a smooth-normal geometry, two lights
    sph = np.concatenate([rng.uniform(-1.5, 1.5, (300, 1)), rng.uniform(0.1, 1.8, (300, 1)),
                          rng.uniform(-1.5, 1.0, (300, 1)), rng.uniform(0.03, 0.2, (300, 1))], 1)
    mats = np.stack([ol.material13(ka=c, kd=c) for c in rng.uniform(0.2, 0.9, (300, 3))])
    third = {"vertex": np.array([[-1.8, 1.6, 1.2], [-1.8, 1.9, 1.2], [-1.5, 1.6, 1.0]], np.float32),
             "face_index": np.array([[0, 1, 2]]), "material": ol.LIGHT_B}
    for geoms, (W, H) in ((base["geometry"], (200, 117)), (base["geometry"] + [third], (97, 61))):
        d = ol.scene_dict([dict(g) for g in geoms], sph.astype(np.float32), mats)
        for index_order in (0, esc.ESC_RENDER_INDEX_ORDER):
            gpu, u8, ref = render_both(esc, renderer, d, (0, 1, 3), (0, 1, 0), W, H,
                                       flags=flags | index_order)
            assert_bit_equal(gpu, ref, f"{form}/{len(d['light_sources'])} lights/{index_order}")
            assert np.array_equal(u8, ol.oracle_quantise(ref))
    # a two-face light with the hashed face choice, triangles only
    d = ol.load_dump("CornellBox-Original")
    gpu, _, ref = render_both(esc, renderer, d, (0, 1, 2), (0, 1, 0), 128, 96,
                              face_mode=esc.ESC_FACE_HASH, seed=5, flags=flags)
    assert_bit_equal(gpu, ref, f"{form}/hashed faces")
    # cyclic strips (the multi-GPU partition) through the same form
    sc, dd = synthetic_dict(esc, "c3", 300)
    eye, look = esc.synthetic_view()
    W, H = 136, 77
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    ref = ol.oracle_render(dd, eye, look, W, H, threads=8)
    world = 3
    rows = multigpu.max_local_rows(H, world)
    gathered = torch.zeros(world, rows * W * 3, dtype=torch.float32, device="cuda:0")
    for rank in range(world):
        renderer.render_strips(cam, W, H, rank, world, out_f32=gathered[rank], flags=flags)
    renderer.synchronize()
    frame = multigpu.assemble_frame_torch(gathered, world, W, H).cpu().numpy()
    assert_bit_equal(frame, ref, f"{form}/strips")


def test_per_kernel_timing_events(esc, renderer):
    """ESC_RENDER_TIME_KERNELS / esc_last_kernel_ms: what bench.py's `kernels` view is built on.
    Both shading forms report two positive durations that add up to about the frame's own time;
    without the flag the call is refused instead of returning stale numbers."""
    import time
    sc = esc.Scene.synthetic("c4", 3000)
    eye, look = esc.synthetic_view()
    W, H = 960, 540
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    ref = renderer.render(cam, W, H)
    for form in (esc.ESC_RENDER_SHADE_QUEUE, esc.ESC_RENDER_SHADE_FUSED):
        renderer.render(cam, W, H, flags=form)  # warm
        renderer.synchronize()
        t0 = time.perf_counter()
        img = renderer.render(cam, W, H, flags=form | esc.ESC_RENDER_TIME_KERNELS)
        wall_ms = (time.perf_counter() - t0) * 1e3
        prim, shade = renderer.last_kernel_ms()
        assert_bit_equal(img, ref, "timed frame")
        assert prim > 0 and shade > 0
        assert prim + shade < wall_ms  # the host call also copies the frame back
    renderer.render(cam, W, H)
    with pytest.raises(esc.EscError):
        renderer.last_kernel_ms()


def test_mixed_triangles_and_spheres(esc, renderer):
    d = ol.load_dump("two")
    rng = np.random.default_rng(5)
    sph = np.concatenate([rng.uniform(-1.5, 1.5, (40, 1)), rng.uniform(0.1, 1.5, (40, 1)),
                          rng.uniform(-1.5, 1.0, (40, 1)), rng.uniform(0.05, 0.3, (40, 1))], 1)
    mats = np.stack([ol.material13(ka=c, kd=c) for c in rng.uniform(0.2, 0.9, (40, 3))])
    d2 = ol.scene_dict(d["geometry"], sph, mats)
    for stage in (esc.ESC_STAGE_SMEM, esc.ESC_STAGE_LDS):
        for px in (1, 2, 4):
            gpu, u8, ref = render_both(esc, renderer, d2, (0, 1, 3), (0, 1, 0), 160, 90,
                                       stage=stage, px=px)
            assert_bit_equal(gpu, ref, f"mixed/px{px}")


def test_heightfield_c5_small(esc, renderer):
    sc, d = synthetic_dict(esc, "c5", 24)  # 1,152 triangles
    eye, look = esc.synthetic_view()
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, 160, 90)
    gpu = renderer.render(cam, 160, 90)
    ref = ol.oracle_render(d, eye, look, 160, 90, threads=8)
    assert_bit_equal(gpu, ref, "c5/24")
    assert ref.sum() > 0


# ---------------------------------------------------------------- reference-pinned twins
@pytest.mark.parametrize("config,n,subdiv,smooth", [("c2", 100, 2, False), ("c3", 1000, 1, True),
                                                    ("c4", 10000, 0, False)])
def test_icosphere_twins_of_the_sphere_configs(esc, renderer, config, n, subdiv, smooth, bvh_tree):
    """each sphere config has a twin made of the reference's only primitive --
    every sphere a tessellated icosahedron (32,003 / 80,003 / 200,003 triangles, one geometry per
    sphere; the c3 twin carries per-vertex normals, quirk S1) -- rendered at 160x90 with the
    config's own semantics through the reference-pinned triangle path: brute force (both stagings)
    and the acceleration structure, each bit-equal to the oracle."""
    sc, d = synthetic_dict(esc, config, n)
    tw = ol.icosphere_twin(d, subdiv, smooth_normals=smooth)
    assert sum(len(g["face_index"]) for g in tw["geometry"]) == 3 + n * 20 * 4 ** subdiv
    eye, look = esc.synthetic_view()
    W, H = 160, 90
    shadows = config != "c2"
    ref, rc = ol.oracle_render_rows(tw, eye, look, W, H, list(range(H)), shadows=shadows,
                                    threads=16, fast=True)
    assert 0 < rc["hit_pixels"] < W * H
    renderer.upload(ol.scene_to_product(tw))
    cam = esc.Camera.for_image(eye, look, W, H)
    # brute force: default order (triangle groups, csrc/rt_device.h TriGroups) and index order,
    # whose any-hit count is the reference's
    for stage, flags in ((esc.ESC_STAGE_SMEM, 0), (esc.ESC_STAGE_SMEM, esc.ESC_RENDER_INDEX_ORDER),
                         (esc.ESC_STAGE_LDS, 0), (esc.ESC_STAGE_BVH, 0)):
        renderer.reset_counters()
        gpu, u8 = renderer.render(cam, W, H, want_u8=True, shadows=shadows, stage=stage, flags=flags)
        cnt = renderer.counters()
        assert_bit_equal(gpu, ref, f"twin/{config}/stage{stage}/flags{flags}")
        assert np.array_equal(u8, ol.oracle_quantise(ref))
        for k in ("primary_rays", "hit_pixels", "shadow_rays"):
            assert cnt[k] == rc[k]
        if stage == esc.ESC_STAGE_LDS or flags == esc.ESC_RENDER_INDEX_ORDER:
            assert cnt["anyhit_tests"] == rc["anyhit_tests"]


def test_cornellbox_water_largest_bundled_mesh(esc, renderer, bvh_tree):
    """CornellBox-Water: 7,088 triangles in 9 geometries, the largest mesh the reference bundles
    (geometry = the reference loader's dump), two-face light with the hashed face choice.  Its
    water material has ks != 0, so device powf vs glibc powf may differ in the last ulp there:
    1e-5 (north_star's tolerance) on those pixels, bit-exact elsewhere; BVH == brute force."""
    d = ol.load_dump("CornellBox-Water")
    assert sum(len(g["face_index"]) for g in d["geometry"]) == 7088
    W, H = 320, 240
    gpu, u8, ref = render_both(esc, renderer, d, (0, 1, 3), (0, 1, 0), W, H,
                               face_mode=esc.ESC_FACE_HASH, seed=3)
    assert np.allclose(gpu, ref, rtol=1e-5, atol=1e-5)
    assert float((bits(gpu) == bits(ref)).mean()) > 0.999
    assert (u8.astype(int) - ol.oracle_quantise(ref).astype(int)).__abs__().max() <= 1
    cam = esc.Camera.for_image((0, 1, 3), (0, 1, 0), W, H)
    bvh = renderer.render(cam, W, H, face_mode=esc.ESC_FACE_HASH, seed=3, stage=esc.ESC_STAGE_BVH)
    assert_bit_equal(bvh, gpu, "Water: BVH vs brute force")
    assert ref.sum() > 0


# ---------------------------------------------------------------- shapes, bands, drop-in
@pytest.mark.parametrize("px", [1, 2, 4])
@pytest.mark.parametrize("W,H", [(33, 9), (97, 61), (130, 75), (2, 2), (64, 8), (31, 7), (128, 8)])
def test_ragged_sizes(esc, renderer, W, H, px):
    """partial tiles in both directions, W % 4 != 0 (byte-store path of the u8 output)"""
    d = ol.load_dump("one")
    gpu, u8, ref = render_both(esc, renderer, d, (0, 1, 3), (0, 1, 0), W, H, px=px)
    assert_bit_equal(gpu, ref, f"{W}x{H}/px{px}")
    assert np.array_equal(u8, ol.oracle_quantise(ref))


def test_row_bands_equal_full_frame(esc, renderer):
    """N-band tiling == 1 band: a pixel is a pure function of (w, h) (main.cpp:628-636)."""
    import torch
    sc, d = synthetic_dict(esc, "c3", 300)
    eye, look = esc.synthetic_view()
    W, H = 200, 117
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    full = renderer.render(cam, W, H)
    for n in (2, 3, 8):
        out = np.zeros_like(full)
        for i in range(n):
            r0, r1 = i * (H // n), (H if i == n - 1 else (i + 1) * (H // n))
            buf = torch.zeros((r1 - r0) * W * 3, dtype=torch.float32, device="cuda:0")
            renderer.render_rows(cam, W, H, r0, r1, out_f32=buf)
            renderer.synchronize()
            out[r0:r1] = buf.cpu().numpy().reshape(r1 - r0, W, 3)
        assert_bit_equal(out, full, f"{n} bands")
    img, u8, ms = esc.render_multi(sc, cam, W, H, 4, want_u8=True)
    assert_bit_equal(img, full, "render_multi")
    assert np.array_equal(u8, esc.quantise(full))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_strips_gather_assemble(esc, renderer, world):
    """the multi-GPU partition on one device: every rank's strips, the padded gather layout,
    the HIP assemble kernel and its torch mirror all reproduce the 1-GPU frame"""
    import torch
    from esctp1raytracer_amd import multigpu
    sc, d = synthetic_dict(esc, "c3", 200)
    eye, look = esc.synthetic_view()
    W, H = 136, 77  # ragged: last strip is 5 rows, W % 32 != 0
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    full, full_u8 = renderer.render(cam, W, H, want_u8=True)
    max_rows = multigpu.max_local_rows(H, world)
    for dtype, px, ref in ((torch.float32, 12, full), (torch.uint8, 3, full_u8)):
        gathered = torch.zeros(world, max_rows * W * 3, dtype=dtype, device="cuda:0")
        for rank in range(world):
            rows = multigpu.local_rows(H, rank, world)
            assert rows == esc.strip_local_rows(H, 8, rank, world)
            kw = {"out_f32": gathered[rank]} if dtype == torch.float32 else {"out_u8": gathered[rank]}
            assert renderer.render_strips(cam, W, H, rank, world, **kw) == rows
        renderer.synchronize()
        frame = torch.zeros(H * W * 3, dtype=dtype, device="cuda:0")
        renderer.assemble_strips(gathered, world, max_rows * W * px, W, H, frame, bytes_per_pixel=px)
        renderer.synchronize()
        got = frame.cpu().numpy().reshape(H, W, 3)
        mirror = multigpu.assemble_frame_torch(gathered, world, W, H).cpu().numpy()
        assert np.array_equal(got, mirror)
        if dtype == torch.float32:
            assert_bit_equal(got, ref, f"strips world={world}")
        else:
            assert np.array_equal(got, ref)


@pytest.mark.parametrize("gather", ["auto", "f32"])
def test_bench_two_ranks_gloo_on_one_gpu(tmp_path, gather):
    """bench.py's N>1 path end to end (strips -> gather -> assemble -> reduced counters) with two
    gloo ranks sharing the one GPU; rank 0 checks rows of the assembled frame against the oracle.
    (Real runs use RCCL, one rank per GPU; this rehearses everything but the transport.)"""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ol.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--config", "c3", "--width", "640", "--height", "360",
           "--verify-rows", "12", "--gather", gather]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["assembled_frame_rows_bit_exact"] is True
    assert ("u8" if gather == "auto" else "fp32") in out["config"]["gather"]
    assert out["config"]["primary_rays_per_frame"] == 640 * 360
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0


@pytest.mark.parametrize("use_rccl", [True, False])
def test_native_multi_gpu_entry_on_one_device(esc, renderer, use_rccl):
    """esc_multi_* / esc_render_frame_multi_rccl: the single-process entry a
    C++ host uses.  One GPU here, so the communicator has one rank and the exchange step has no
    peer -- what runs is dlopen(librccl) + ncclCommInitAll/Destroy, the strip render, the padded
    gather layout, k_assemble_strips and the copy back; the N-rank partition itself is the
    esc_render_strips one (test_strips_gather_assemble covers worlds 1/2/3/8)."""
    sc, d = synthetic_dict(esc, "c3", 300)
    eye, look = esc.synthetic_view()
    W, H = 200, 117  # ragged last strip
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    full, full_u8 = renderer.render(cam, W, H, want_u8=True)
    if use_rccl:
        assert esc.rccl_available()
    m = esc.MultiRenderer(1, use_rccl=use_rccl)
    m.upload(sc)
    img, ms, dptr = m.render(cam, W, H)
    assert_bit_equal(img, full, "esc_multi_render fp32")
    assert ms[0] > 0 and dptr
    u8, _, _ = m.render(cam, W, H, gather_u8=True)
    assert np.array_equal(u8, full_u8)
    bvh, _, _ = m.render(cam, W, H, stage=esc.ESC_STAGE_BVH)
    assert_bit_equal(bvh, full, "esc_multi_render through the BVH")
    m.close()
    if use_rccl:
        with pytest.raises(esc.EscError):  # one communicator rank per device: 2 > device count here
            esc.MultiRenderer(2, use_rccl=True)
        with pytest.raises(esc.EscError):  # ... and distinct ones
            esc.MultiRenderer(2, device_ids=[0, 0], use_rccl=True)
    with pytest.raises(esc.EscError):
        esc.MultiRenderer(1, device_ids=[5], use_rccl=use_rccl)
    if use_rccl:
        img2, u82, ms2 = esc.render_multi_rccl(sc, cam, W, H, 1, want_u8=True)
        assert_bit_equal(img2, full, "esc_render_frame_multi_rccl")
        assert np.array_equal(u82, full_u8)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_native_multi_gpu_n_ranks_sharing_one_device(esc, renderer, n):
    """The n > 1 branch of esc_multi_render (csrc/rt_multi.cpp): without RCCL ranks may share a
    device, so the whole n-rank path -- one context and stream per rank, round-robin 8-row strips,
    the per-rank pitch of the padded gather layout, the copies into device 0's blocks ordered by
    events, k_assemble_strips, the copy back, per-rank timings -- runs on the one GPU of the test
    box.  fp32 and u8 gathers, brute force and BVH, each equal to the 1-rank frame and the oracle.
    (The RCCL exchange itself needs n distinct devices and stays unverified: INTEGRATION.md.)"""
    sc, d = synthetic_dict(esc, "c3", 300)
    eye, look = esc.synthetic_view()
    W, H = 200, 117  # ragged last strip; at n = 8 some ranks get one strip less
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    full, full_u8 = renderer.render(cam, W, H, want_u8=True)
    ref = ol.oracle_render(d, eye, look, W, H, threads=8)
    assert_bit_equal(full, ref, "1-rank frame")
    for ids in (None, [0] * n):
        m = esc.MultiRenderer(n, device_ids=ids, use_rccl=False)
        m.upload(sc)
        for _ in range(2):  # the second frame reuses every buffer and cached list
            img, ms, dptr = m.render(cam, W, H)
            assert_bit_equal(img, ref, f"esc_multi_render fp32, {n} ranks")
            assert dptr and all(x > 0 for x in ms), ms  # every rank reports its own render time
        u8, _, _ = m.render(cam, W, H, gather_u8=True)
        assert np.array_equal(u8, full_u8)
        bvh, _, _ = m.render(cam, W, H, stage=esc.ESC_STAGE_BVH)
        assert_bit_equal(bvh, ref, f"esc_multi_render through the BVH, {n} ranks")
        small, _, _ = m.render(esc.Camera.for_image(eye, look, 64, 9), 64, 9)  # fewer strips than ranks
        assert_bit_equal(small, ol.oracle_render(d, eye, look, 64, 9, threads=4), f"64x9, {n} ranks")
        m.close()


def _oracle_scene_in_flat_order(flat):
    """The oracle's scene with the triangles in the order of a flattened scene's triangles[] (one
    geometry per triangle, its material and normals as the flat triangle carries them) and the
    lights in the order of lights[]: what `trace` must render for these arrays.  One-face lights only
    (a light geometry has to stay one geometry for quirk S2 / the light count)."""
    tris = [flat.triangles[i] for i in range(flat.num_triangles)]
    geoms = []
    for t in tris:
        v = np.array([[t.vertices[k][c] for c in range(3)] for k in range(3)], np.float32)
        g = {"vertex": v, "face_index": np.array([[0, 1, 2]]),
             "material": ol.material13(ka=list(t.ka), kd=list(t.kd), ks=list(t.ks), ke=list(t.ke), Ns=t.Ns)}
        if t.has_normals:
            g["normals"] = np.array([[t.normals[k][c] for c in range(3)] for k in range(3)], np.float32)
        geoms.append(g)
    d = ol.scene_dict(geoms)
    order = []
    for L in range(flat.num_lights):
        lt = flat.lights[L]
        assert lt.num_light_faces == 1
        face = flat.light_triangles[lt.light_faces[0]]
        want = [[face.vertices[k][c] for c in range(3)] for k in range(3)]
        pos = [i for i, t in enumerate(tris)
               if t.is_light and [[t.vertices[k][c] for c in range(3)] for k in range(3)] == want]
        assert len(pos) == 1
        order.append(pos[0])
    assert sorted(order) == sorted(d["light_sources"])
    d["light_sources"] = order  # the light LOOP order is the scene's, not the sort's (quirk S3)
    return d


def test_trace_drop_in(esc, renderer):
    """The ispc::trace symbol on FlatScene arrays (trace.ispc:86-92 / main.cpp:619-624).  In (geometry,
    face) order it is the scene path's image.  With the reference's centroid-x sort
    (flatten_iscp.cpp:14-21,110) the primitive order changes -- equal-t ties and, with two lights,
    the first occluder whose t2 moves the next light's ray (quirk S3) follow it -- so the sorted
    arrays are rendered through the ORACLE in that same order and must match bit for bit: scene
    `two` (two lights, per-vertex normals) and the Cornell box with its light cut to one triangle."""
    d = ol.load_dump("CornellBox-Original")
    sc = ol.scene_to_product(d)
    W, H = 128, 96
    aspect = np.float32(W) / np.float32(H)
    ref = ol.oracle_render(d, (0, 1, 2), (0, 1, 0), W, H, face_mode=ol.ORC_FACE_HASH, seed=0)
    flat = sc.flatten_ispc(sort_by_centroid_x=False)
    assert flat.num_triangles == 36 and flat.num_lights == 1 and flat.num_light_triangles == 2
    img = esc.trace(W, H, (0, 1, 2), (0, 1, 0), (0, 1, 0), 60.0, aspect, flat)
    assert_bit_equal(img, ref, "trace()")
    import os
    os.environ["ESC_TRACE_STAGE"] = "bvh"  # the seam's only way to opt into the tree
    try:
        img3 = esc.trace(W, H, (0, 1, 2), (0, 1, 0), (0, 1, 0), 60.0, aspect, flat)
    finally:
        del os.environ["ESC_TRACE_STAGE"]
    assert_bit_equal(img3, ref, "trace() through the BVH")
    # ---- the reference's sort, pinned through the oracle in the permuted order
    cornell1 = ol.load_dump("CornellBox-Original")
    lg = cornell1["geometry"][cornell1["light_sources"][0]]
    lg["face_index"] = lg["face_index"][:1]  # one face: the light sample is deterministic (S2 / S8)
    lg["vertex"] = lg["vertex"][:3]
    cornell1 = ol.scene_dict(cornell1["geometry"])
    for name, dd, eye in (("two", ol.load_dump("two"), (0, 1, 3)), ("cornell, 1-face light", cornell1, (0, 1, 2))):
        scd = ol.scene_to_product(dd)
        unsorted = scd.flatten_ispc(sort_by_centroid_x=False)
        srt = scd.flatten_ispc(sort_by_centroid_x=True)
        order_u = [tuple(unsorted.triangles[i].vertices[0]) for i in range(unsorted.num_triangles)]
        order_s = [tuple(srt.triangles[i].vertices[0]) for i in range(srt.num_triangles)]
        assert sorted(order_u) == sorted(order_s) and order_u != order_s, name + ": the sort permutes"
        base = ol.oracle_render(dd, eye, (0, 1, 0), W, H)
        assert_bit_equal(esc.trace(W, H, eye, (0, 1, 0), (0, 1, 0), 60.0, aspect, unsorted), base,
                         f"trace()/{name}/unsorted")
        want = ol.oracle_render(_oracle_scene_in_flat_order(srt), eye, (0, 1, 0), W, H)
        got = esc.trace(W, H, eye, (0, 1, 0), (0, 1, 0), 60.0, aspect, srt)
        assert_bit_equal(got, want, f"trace()/{name}/sorted by centroid x")


def test_empty_and_missing(esc, renderer):
    """no primitives hit / no lights: black frame; zero-row band is a no-op"""
    sc = esc.Scene()
    v = np.array([[0, 0, -100], [1, 0, -100], [0, 1, -100]], np.float32)
    sc.add_geometry(v, [[0, 1, 2]], ol.material13(ka=(1, 1, 1), kd=(1, 1, 1)))
    renderer.upload(sc)
    cam = esc.Camera.for_image((0, 0, 5), (0, 0, 6), 64, 48)  # looking away
    img = renderer.render(cam, 64, 48)
    assert not img.any()
    renderer.render_rows(cam, 64, 48, 10, 10)  # empty band
    with pytest.raises(esc.EscError):
        renderer.render_rows(cam, 64, 48, 40, 50)  # beyond H
    with pytest.raises(esc.EscError):
        renderer.render(cam, 1, 48)  # W - 1 == 0 divides at main.cpp:709


# ---------------------------------------------------------------- BASELINE.json full size
def test_c4_full_size_properties(esc, renderer):
    """3840x2160 / 10k spheres: (a) sampled rows bit-equal to the oracle, (b) two bands ==
    one frame, (c) counters add up, (d) u8 == quantise(fp32)."""
    import torch
    sc, d = synthetic_dict(esc, "c4", 10000)
    eye, look = esc.synthetic_view()
    W, H = 3840, 2160
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    f32 = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
    u8 = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda:0")
    renderer.reset_counters()
    renderer.render_rows(cam, W, H, 0, H, out_f32=f32, out_u8=u8)
    cnt = renderer.counters()
    full = f32.cpu().numpy().reshape(H, W, 3)
    assert cnt["primary_rays"] == W * H
    assert cnt["shadow_rays"] == cnt["hit_pixels"]  # one light, one shadow ray per hit pixel
    assert 0 < cnt["hit_pixels"] < W * H
    assert np.array_equal(u8.cpu().numpy().reshape(H, W, 3), esc.quantise(full))
    osc = ol.OracleScene(d)
    for r in (0, 700, 1079, 1080, 1500, 2159):
        ref = ol.oracle_render(osc, eye, look, W, H, rows=(r, r + 1), threads=16)
        assert_bit_equal(full[r], ref[r], f"row {r}")
    half = torch.zeros((H // 2) * W * 3, dtype=torch.float32, device="cuda:0")
    for r0 in (0, H // 2):
        renderer.render_rows(cam, W, H, r0, r0 + H // 2, out_f32=half)
        renderer.synchronize()
        assert_bit_equal(half.cpu().numpy().reshape(H // 2, W, 3), full[r0:r0 + H // 2], "band")


@pytest.mark.parametrize("config,W,H,rows", [
    ("c2", 1920, 1080, (0, 270, 539, 540, 800, 1079)),
    ("c3", 3840, 2160, (0, 700, 1079, 1080, 1500, 2159)),
    ("c5", 7680, 4320, (0, 1400, 2160, 3000, 4319))])
def test_full_size_rows_vs_oracle(esc, renderer, config, W, H, rows):
    """Every BASELINE config at its OWN size against the oracle (c4: the test above): sampled rows
    of the full frame, brute force, bit for bit -- fp32 and quantised bytes -- plus the counters'
    size-independent identities.  c5 = 7680x4320 / 100,352-triangle heightfield."""
    import torch
    sc = esc.Scene.synthetic(config)
    d = ol.scene_from_product(sc)
    eye, look = esc.synthetic_view()
    shadows = config != "c2"
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    f32 = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
    u8 = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda:0")
    renderer.reset_counters()
    renderer.render_rows(cam, W, H, 0, H, out_f32=f32, out_u8=u8, shadows=shadows)
    cnt = renderer.counters()
    assert cnt["primary_rays"] == W * H
    assert cnt["shadow_rays"] == (cnt["hit_pixels"] if shadows else 0)
    assert 0 < cnt["hit_pixels"] <= W * H
    rr = list(rows)
    ref, rc = ol.oracle_render_rows(d, eye, look, W, H, rr, shadows=shadows, threads=16, fast=True)
    idx = torch.tensor(rr, device="cuda:0")
    got = f32.view(H, W, 3)[idx].cpu().numpy()
    got8 = u8.view(H, W, 3)[idx].cpu().numpy()
    for i, r in enumerate(rr):
        assert_bit_equal(got[i], ref[i], f"{config} row {r}")
    assert np.array_equal(got8, ol.oracle_quantise(ref))
    assert ref.sum() > 0
    del f32, u8


# ---------------------------------------------------------------- ESC_STAGE_BVH
# The tree only decides which primitives get the exact tests; the image must be the brute-force
# image.  Small scenes are checked against the oracle, full-size frames against the brute-force
# kernels on the same GPU (every pixel, both the fp32 frame and the quantised bytes).
@pytest.mark.parametrize("name,eye", [("one", (0, 1, 3)), ("two", (0, 1, 3)),
                                      ("CornellBox-Original", (0, 1, 2)),
                                      ("CornellBox-Empty-CO", (0, 1, 3)),
                                      ("cornell_box", (0, 1, 3))])
def test_bvh_triangle_scenes_vs_oracle(esc, renderer, name, eye, bvh_tree):
    d = ol.load_dump(name)
    gpu, u8, ref = render_both(esc, renderer, d, eye, (0, 1, 0), 160, 90,
                               stage=esc.ESC_STAGE_BVH)
    assert_bit_equal(gpu, ref, f"bvh/{name}")
    assert np.array_equal(u8, ol.oracle_quantise(ref))
    info = renderer.accel_info()
    assert info["builds"] >= 1 and info["tri_blocks"] > 0


@pytest.mark.parametrize("config,n,shadows", [("c2", 100, False), ("c3", 1000, True),
                                              ("c4", 613, True), ("c4", 10000, True),
                                              ("c5", 24, True), ("c2", 1, True), ("c2", 5, True)])
def test_bvh_synthetic_scenes_vs_oracle(esc, renderer, config, n, shadows):
    sc, d = synthetic_dict(esc, config, n)
    eye, look = esc.synthetic_view()
    W, H = (192, 108) if n < 5000 else (96, 54)
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    renderer.reset_counters()
    gpu, u8 = renderer.render(cam, W, H, want_u8=True, shadows=shadows, stage=esc.ESC_STAGE_BVH)
    cnt = renderer.counters()
    ref, rc = ol.oracle_render(d, eye, look, W, H, shadows=shadows, threads=8,
                               return_counters=True)
    assert_bit_equal(gpu, ref, f"bvh/{config}/{n}")
    assert np.array_equal(u8, ol.oracle_quantise(ref))
    for k in ("primary_rays", "hit_pixels", "shadow_rays"):
        assert cnt[k] == rc[k]
    if shadows and n >= 100:  # the tree must leave fewer tests than the reference's linear scan
        assert cnt["anyhit_tests"] < rc["anyhit_tests"]


def test_bvh_two_lights_first_occluder_in_order(esc, renderer, bvh_tree):
    """Quirk S3 carries the FIRST occluder's t2 into the next light's shadow ray: with two
    lights and many overlapping occluders the walk must report the same one as the linear scan."""
    d = ol.load_dump("two")
    rng = np.random.default_rng(11)
    sph = np.concatenate([rng.uniform(-1.5, 1.5, (120, 1)), rng.uniform(0.1, 1.8, (120, 1)),
                          rng.uniform(-1.5, 1.0, (120, 1)), rng.uniform(0.05, 0.35, (120, 1))], 1)
    mats = np.stack([ol.material13(ka=c, kd=c) for c in rng.uniform(0.2, 0.9, (120, 3))])
    d2 = ol.scene_dict(d["geometry"], sph, mats)
    assert len(d2["light_sources"]) >= 2
    for fm, eye in ((esc.ESC_FACE_FIXED, (0, 1, 3)), (esc.ESC_FACE_HASH, (0.3, 1.4, 2.5))):
        gpu, u8, ref = render_both(esc, renderer, d2, eye, (0, 1, 0), 200, 120,
                                   stage=esc.ESC_STAGE_BVH, face_mode=fm, fixed_face=0, seed=7)
        assert_bit_equal(gpu, ref, f"bvh/two-lights/{fm}")
        brute = renderer.render(esc.Camera.for_image(eye, (0, 1, 0), 200, 120), 200, 120,
                                face_mode=fm, fixed_face=0, seed=7)
        assert_bit_equal(gpu, brute, f"bvh vs brute/two-lights/{fm}")
    # the lights of this scene have one face each: face 1 does not exist (main.cpp:743-748)
    with pytest.raises(esc.EscError):
        renderer.render(esc.Camera.for_image((0, 1, 3), (0, 1, 0), 64, 48), 64, 48,
                        face_mode=esc.ESC_FACE_FIXED, fixed_face=1)


@pytest.mark.parametrize("elev", [1e-2, 1e-3, 1e-4, 1e-5])
@pytest.mark.parametrize("dist", [30.0, 300.0])
def test_bvh_grazing_rays_on_large_triangles(esc, renderer, elev, dist, bvh_tree):
    """Rays that graze large triangles' planes from far away: the camera sits
    `dist` away from a 24 x 28 floor, `elev` radians above its plane, and looks along it, so every
    primary ray meets the floor plane at <= ~elev + fov/2 ... down to ~elev, and the image's
    lower rows cross the floor's far and side edges at grazing incidence -- where the reference
    arithmetic's (u, v) are dominated by rounding.  Walls and a ceiling graze the other way.
    ESC_STAGE_BVH (tree walk and bins) must still return brute force's frame, which is the
    oracle's."""
    fl = np.array([[-12, 0, 4], [12, 0, 4], [12, 0, -24], [-12, 0, 4], [12, 0, -24], [-12, 0, -24]], np.float32)
    wall = np.array([[-12, 0, 4], [-12, 0, -24], [-12, 9, -24], [-12, 0, 4], [-12, 9, -24], [-12, 9, 4]], np.float32)
    ceil = fl[::-1].copy()
    ceil[:, 1] = 9.0
    light = np.array([[-0.5, 8.9, -9.5], [0.0, 8.9, -10.5], [0.5, 8.9, -9.5]], np.float32)
    geoms = [{"vertex": fl, "face_index": np.arange(6).reshape(2, 3), "material": ol.WHITE},
             {"vertex": wall, "face_index": np.arange(6).reshape(2, 3), "material": ol.RED},
             {"vertex": ceil, "face_index": np.arange(6).reshape(2, 3), "material": ol.BLUE},
             {"vertex": light, "face_index": np.array([[0, 1, 2]]), "material": ol.LIGHT_A}]
    rng = np.random.default_rng(5)
    sph = np.concatenate([rng.uniform(-10, 10, (40, 1)), rng.uniform(0.3, 6, (40, 1)),
                          rng.uniform(-22, 2, (40, 1)), rng.uniform(0.2, 0.8, (40, 1))], 1)
    mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.2, 0.9, (40, 3))])
    d = ol.scene_dict(geoms, sph.astype(np.float32), mats)
    # the camera looks along -z, slightly above the floor plane, aimed at the floor's far edge
    eye = (0.3, float(dist * elev), 4.0 + dist)
    look = (0.3, 0.0, -24.0)
    W, H = 256, 144
    ref = ol.oracle_render(d, eye, look, W, H, threads=8)
    for stage in (esc.ESC_STAGE_AUTO, esc.ESC_STAGE_BVH):
        gpu, u8, _ = render_both(esc, renderer, d, eye, look, W, H, stage=stage)
        assert_bit_equal(gpu, ref, f"grazing elev={elev} dist={dist} stage={stage}")


def test_bvh_second_light_origin_outside_the_scene_box(esc, renderer, bvh_tree):
    """Quirk S3 with two lights: light 2's shadow ray starts at camera + dir * (t_occ - eps), where
    t_occ is light 1's occluder distance ALONG ITS SHADOW RAY.  With light 1 far away and its
    occluder next to it, t_occ is several times the primary hit distance: the origin lies far
    past the floor, outside the box of the scene.  The box pads of the tree must hold there too
    (OriginBounds now includes a camera-centred ball when there is more than one light
    point).  Spheres sit under the floor, where those origins land, on the way to light 2."""
    rng = np.random.default_rng(21)
    floor = np.array([[-6, 0, 4], [6, 0, 4], [6, 0, -6], [-6, 0, 4], [6, 0, -6], [-6, 0, -6]], np.float32)
    lA = np.array([[-0.5, 30, -40], [0.5, 30, -40], [0, 30, -41]], np.float32)   # far away
    lB = np.array([[8, -3, 2], [8.4, -3, 2], [8, -2.6, 2.2]], np.float32)          # below the floor level
    geoms = [{"vertex": floor, "face_index": np.arange(6).reshape(2, 3), "material": ol.WHITE},
             {"vertex": lA, "face_index": np.array([[0, 1, 2]]), "material": ol.LIGHT_A},
             {"vertex": lB, "face_index": np.array([[0, 1, 2]]), "material": ol.LIGHT_B}]
    big = np.array([[0.0, 26.0, -35.0, 6.0]])  # hides light A from most of the floor, next to it
    n = 150
    under = np.concatenate([rng.uniform(-8, 8, (n, 1)), rng.uniform(-30, -1, (n, 1)),
                            rng.uniform(-45, 3, (n, 1)), rng.uniform(0.2, 1.5, (n, 1))], axis=1)
    sph = np.concatenate([big, under]).astype(np.float32)
    mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.2, 0.9, (len(sph), 3))])
    d = ol.scene_dict(geoms, sph, mats)
    assert len(d["light_sources"]) == 2
    eye, look = (0.0, 1.0, 3.0), (0.0, 0.0, 0.0)
    W, H = 224, 128
    ref, rc = ol.oracle_render(d, eye, look, W, H, threads=8, return_counters=True)
    assert rc["hit_pixels"] > W * H // 4
    for stage in (esc.ESC_STAGE_AUTO, esc.ESC_STAGE_BVH):
        gpu, u8, _ = render_both(esc, renderer, d, eye, look, W, H, stage=stage)
        assert_bit_equal(gpu, ref, f"S3 far origin/stage{stage}")
    assert ref.sum() > 0


def test_bvh_third_light_origin_two_diagonals_out(esc, renderer, bvh_tree):
    """Quirk S3 with THREE lights: light 3's shadow ray starts at camera + dir * t where
    t was left by light 2's ray, which itself started far outside the scene and is therefore up to
    two scene diagonals long: OriginBounds' ball is (light points - 1) diagonals.  Light A far away
    with its occluder next to it pushes light B's origins out; light B on the opposite far side with
    ITS occluder next to it pushes light C's origins further still; spheres sit out there, on the
    way to light C.  Brute force (lists, sweep) and the BVH stage against the oracle."""
    rng = np.random.default_rng(33)
    floor = np.array([[-6, 0, 4], [6, 0, 4], [6, 0, -6], [-6, 0, 4], [6, 0, -6], [-6, 0, -6]], np.float32)
    lA = np.array([[-0.5, 30, -40], [0.5, 30, -40], [0, 30, -41]], np.float32)
    lB = np.array([[-0.5, 40, 60], [0.5, 40, 60], [0, 40, 61]], np.float32)
    lC = np.array([[8, -3, 2], [8.4, -3, 2], [8, -2.6, 2.2]], np.float32)
    geoms = [{"vertex": floor, "face_index": np.arange(6).reshape(2, 3), "material": ol.WHITE}]
    for lt, m in ((lA, ol.LIGHT_A), (lB, ol.LIGHT_B), (lC, ol.LIGHT_A)):
        geoms.append({"vertex": lt, "face_index": np.array([[0, 1, 2]]), "material": m})
    big = np.array([[0.0, 26.0, -35.0, 6.0], [0.0, 36.0, 54.0, 7.0]])  # next to A and next to B
    n = 200
    out = np.concatenate([rng.uniform(-12, 12, (n, 1)), rng.uniform(-60, -1, (n, 1)),
                          rng.uniform(-80, 20, (n, 1)), rng.uniform(0.3, 2.5, (n, 1))], axis=1)
    sph = np.concatenate([big, out]).astype(np.float32)
    mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.2, 0.9, (len(sph), 3))])
    d = ol.scene_dict(geoms, sph, mats)
    assert len(d["light_sources"]) == 3
    eye, look = (0.0, 1.0, 3.0), (0.0, 0.0, 0.0)
    W, H = 224, 128
    ref, rc = ol.oracle_render(d, eye, look, W, H, threads=8, return_counters=True)
    undo_s3 = ol.oracle_render(d, eye, look, W, H, threads=8, quirks=ol.ORC_QUIRK_S1)
    assert (bits(ref) != bits(undo_s3)).sum() > 1000, "the scene does not exercise quirk S3"
    for stage, flags in ((esc.ESC_STAGE_AUTO, 0),
                         (esc.ESC_STAGE_AUTO, esc.ESC_RENDER_NO_TILE_LISTS | esc.ESC_RENDER_NO_LIGHT_LISTS),
                         (esc.ESC_STAGE_BVH, 0)):
        gpu, u8, _ = render_both(esc, renderer, d, eye, look, W, H, stage=stage, flags=flags)
        assert_bit_equal(gpu, ref, f"S3 three lights/stage{stage}/flags{flags}")
    assert ref.sum() > 0


@pytest.mark.parametrize("W,H", [(33, 9), (97, 61), (2, 2), (31, 7)])
def test_bvh_ragged_sizes(esc, renderer, W, H, bvh_tree):
    d = ol.load_dump("one")
    gpu, u8, ref = render_both(esc, renderer, d, (0, 1, 3), (0, 1, 0), W, H,
                               stage=esc.ESC_STAGE_BVH)
    assert_bit_equal(gpu, ref, f"bvh/{W}x{H}")


def test_bvh_multi_band_single_process(esc, bvh_tree):
    """esc_render_frame_multi: every band's context builds its own tree"""
    sc, d = synthetic_dict(esc, "c3", 400)
    eye, look = esc.synthetic_view()
    cam = esc.Camera.for_image(eye, look, 200, 120)
    img, _, _ = esc.render_multi(sc, cam, 200, 120, 3, stage=esc.ESC_STAGE_BVH)
    ref = ol.oracle_render(d, eye, look, 200, 120, threads=8)
    assert_bit_equal(img, ref, "bvh/render_multi")


def test_bvh_camera_move_rebuilds(esc, renderer, bvh_tree):
    """the box pads depend on where rays can start; leaving that region must rebuild"""
    sc, d = synthetic_dict(esc, "c3", 200)
    renderer.upload(sc)
    eye, look = esc.synthetic_view()
    before = renderer.accel_info()["builds"]
    for e in (eye, (0.5, 3.2, 6.5), (300.0, 40.0, 250.0)):
        cam = esc.Camera.for_image(e, look, 96, 54)
        gpu = renderer.render(cam, 96, 54, stage=esc.ESC_STAGE_BVH)
        ref = ol.oracle_render(d, e, look, 96, 54, threads=8)
        assert_bit_equal(gpu, ref, f"bvh/eye{e}")
    # first frame builds, the small move stays inside the bounds, the far one rebuilds
    assert renderer.accel_info()["builds"] - before == 2


@pytest.mark.parametrize("config,W,H,bins", [("c2", 1920, 1080, True), ("c3", 3840, 2160, True),
                                             ("c4", 3840, 2160, True), ("c5", 7680, 4320, True),
                                             ("c4", 3840, 2160, False),
                                             ("c5", 7680, 4320, False)])
def test_bvh_full_size_equals_brute_force(esc, renderer, config, W, H, bins, monkeypatch, bvh_tree):
    """BASELINE.json's sizes: every pixel of the BVH frame == the brute-force frame, with the
    screen / light bins and with the plain tree walk (ESC_BVH_BINS=0)."""
    import torch
    if not bins:
        monkeypatch.setenv("ESC_BVH_BINS", "0")
    sc = esc.Scene.synthetic(config)
    eye, look = esc.synthetic_view()
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    shadows = config != "c2"
    a = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
    b = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
    a8 = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda:0")
    b8 = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda:0")
    renderer.reset_counters()
    renderer.render_rows(cam, W, H, 0, H, out_f32=a, out_u8=a8, shadows=shadows)
    ca = renderer.counters()
    renderer.reset_counters()
    renderer.render_rows(cam, W, H, 0, H, out_f32=b, out_u8=b8, shadows=shadows,
                         stage=esc.ESC_STAGE_BVH)
    cb = renderer.counters()
    renderer.synchronize()
    nd = int((a.view(torch.int32) != b.view(torch.int32)).sum().item())
    assert nd == 0, f"{config}: {nd} fp32 values differ between BVH and brute force"
    assert bool((a8 == b8).all().item())
    for k in ("primary_rays", "hit_pixels", "shadow_rays"):
        assert ca[k] == cb[k]
    assert float(a.sum().item()) > 0


@pytest.mark.parametrize("eye,look", [((0.0, 2.5, -9.0), (3.0, 2.0, -15.0)),   # inside the cloud
                                      ((0.0, 2.5, -9.0), (0.0, 2.5, 5.0)),     # looking back out
                                      ((7.9, 0.6, -2.1), (-8.0, 4.0, -20.0)),  # from a corner
                                      ((0.0, 30.0, -10.0), (0.0, 0.0, -10.01)),  # straight down
                                      ((0.0, 3.0, 40.0), (0.0, 3.0, 80.0))])   # everything behind
def test_bvh_bins_awkward_cameras(esc, renderer, eye, look):
    """screen-space bins of the primary pass: primitives behind the camera, straddling the camera
    plane (global list), off screen; every case against the oracle"""
    sc, d = synthetic_dict(esc, "c3", 500)
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, 224, 136)
    gpu = renderer.render(cam, 224, 136, stage=esc.ESC_STAGE_BVH)
    ref = ol.oracle_render(d, eye, look, 224, 136, threads=8)
    assert_bit_equal(gpu, ref, f"bvh bins/eye{eye}")


def test_bvh_bin_overflow_falls_back_to_the_tree(esc, renderer):
    """> 32 spheres behind one another in one tile, and > 16 spheres around the camera: the bin
    (or the global list) overflows and those tiles (or all) walk the tree instead"""
    rng = np.random.default_rng(3)
    n = 90
    line = np.stack([np.full(n, 0.02), np.full(n, 1.0), -np.linspace(2.0, 40.0, n),
                     np.linspace(0.05, 0.6, n)], axis=1)
    ang = np.linspace(0.0, 2 * np.pi, 24, endpoint=False)  # a ring in the camera plane z = 3
    around = np.stack([3.0 * np.cos(ang), 1.0 + 3.0 * np.sin(ang), np.full(24, 3.0),
                       np.full(24, 0.8)], axis=1)
    d = ol.load_dump("one")
    for sph in (line, np.concatenate([line, around])):
        mats = np.stack([ol.material13(ka=c, kd=c) for c in rng.uniform(0.2, 0.9, (len(sph), 3))])
        d2 = ol.scene_dict(d["geometry"], sph.astype(np.float32), mats)
        gpu, u8, ref = render_both(esc, renderer, d2, (0, 1, 3), (0, 1, 0), 160, 96,
                                   stage=esc.ESC_STAGE_BVH)
        assert_bit_equal(gpu, ref, f"bvh bins overflow/{len(sph)}")
        assert ref.sum() > 0


def test_bvh_unaligned_band_and_bins_off(esc, renderer, monkeypatch):
    """a band that starts off the 8-row grid cannot use the bins (tree walk), and ESC_BVH_BINS=0
    switches them off altogether: same pixels either way"""
    import torch
    sc, d = synthetic_dict(esc, "c3", 300)
    eye, look = esc.synthetic_view()
    W, H = 160, 96
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    full = renderer.render(cam, W, H, stage=esc.ESC_STAGE_BVH)
    ref = ol.oracle_render(d, eye, look, W, H, threads=8)
    assert_bit_equal(full, ref, "bvh/full")
    band = torch.zeros(40 * W * 3, dtype=torch.float32, device="cuda:0")
    renderer.render_rows(cam, W, H, 13, 53, out_f32=band, stage=esc.ESC_STAGE_BVH)
    renderer.synchronize()
    assert_bit_equal(band.cpu().numpy().reshape(40, W, 3), ref[13:53], "bvh/unaligned band")
    monkeypatch.setenv("ESC_BVH_BINS", "0")
    nobins = renderer.render(cam, W, H, stage=esc.ESC_STAGE_BVH)
    assert_bit_equal(nobins, ref, "bvh/bins off")


def _spheres_with_lights(esc, light_tris, n=400):
    """c3's spheres + floor, with the given light triangles (one geometry each) instead of c3's"""
    sc, d = synthetic_dict(esc, "c3", n)
    geoms = [g for i, g in enumerate(d["geometry"]) if i not in d["light_sources"]]
    for tri in light_tris:
        geoms.append({"vertex": np.array(tri, np.float32), "face_index": np.array([[0, 1, 2]]),
                      "material": ol.material13(ka=(.78,) * 3, kd=(.78,) * 3, ke=(17, 12, 4))})
    return ol.scene_dict(geoms, d["spheres"], d["sphere_materials"])


@pytest.mark.parametrize("case", ["light inside the cloud", "two lights", "five lights"])
def test_bvh_light_bins_awkward_lights(esc, renderer, case):
    """light-space bins of the shadow pass: a light in the middle of the spheres (boxes straddle
    every cube face: face lists overflow, rays walk the tree), two lights (first-occluder mode
    through the bins), five light points (more than get a cube map: tree walk only)"""
    up = [(-0.5, 12, -9.5), (0.0, 12, -10.5), (0.5, 12, -9.5)]
    mid = [(-0.2, 2.6, -9.8), (0.0, 2.6, -10.2), (0.2, 2.6, -9.8)]
    side = [(-7.5, 6.0, -3.0), (-7.5, 6.5, -3.5), (-7.0, 6.0, -3.0)]
    tris = {"light inside the cloud": [mid], "two lights": [up, side],
            "five lights": [up, side, mid, [(6, 7, -15), (6, 7.5, -15), (6.5, 7, -15)],
                            [(0, 9, -20), (0.5, 9, -20), (0, 9, -20.5)]]}[case]
    d = _spheres_with_lights(esc, tris)
    eye, look = esc.synthetic_view()
    gpu, u8, ref = render_both(esc, renderer, d, eye, look, 224, 128, stage=esc.ESC_STAGE_BVH)
    assert_bit_equal(gpu, ref, f"bvh light bins/{case}")
    assert ref.sum() > 0


@pytest.mark.parametrize("seed", list(range(16)))
def test_bvh_random_scenes(esc, renderer, seed, bvh_tree):
    """random triangle soup + spheres + 1..3 one-face lights, random camera inside or outside:
    brute force and the BVH stage (tree, screen bins, light bins) both equal the oracle"""
    rng = np.random.default_rng(1000 + seed)
    n_tri = int(rng.integers(1, 400))
    n_sph = int(rng.integers(0, 300))
    c = rng.uniform(-4, 4, (n_tri, 1, 3))
    tri = (c + rng.normal(0, rng.uniform(0.05, 1.5), (n_tri, 3, 3))).astype(np.float32)
    geoms = []
    per = max(1, n_tri // 5)
    for k in range(0, n_tri, per):  # a few geometries, one material each
        t = tri[k:k + per].reshape(-1, 3)
        col = rng.uniform(0.1, 0.9, 3)
        geoms.append({"vertex": t, "face_index": np.arange(len(t)).reshape(-1, 3),
                      "material": ol.material13(ka=col, kd=col)})
    for _ in range(int(rng.integers(1, 4))):
        p0 = rng.uniform(-5, 5, 3) + np.array([0, 6, 0])
        lt = np.stack([p0, p0 + rng.normal(0, 0.3, 3), p0 + rng.normal(0, 0.3, 3)])
        geoms.append({"vertex": lt.astype(np.float32), "face_index": np.array([[0, 1, 2]]),
                      "material": ol.material13(ka=(.5,) * 3, kd=(.5,) * 3, ke=(9, 8, 7))})
    sph = np.concatenate([rng.uniform(-4, 4, (n_sph, 3)), rng.uniform(0.03, 0.8, (n_sph, 1))], 1)
    mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.1, 0.9, (max(n_sph, 1), 3))])
    d = ol.scene_dict(geoms, sph.astype(np.float32), mats[:n_sph])
    eye = tuple(float(x) for x in rng.uniform(-6, 6, 3))
    look = tuple(float(x) for x in rng.uniform(-2, 2, 3))
    W, H = 168, 104
    sc = ol.scene_to_product(d)
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    ref = ol.oracle_render(d, eye, look, W, H, threads=8)
    brute = renderer.render(cam, W, H)
    assert_bit_equal(brute, ref, f"random/{seed}/brute")
    bvh = renderer.render(cam, W, H, stage=esc.ESC_STAGE_BVH)
    assert_bit_equal(bvh, ref, f"random/{seed}/bvh")


def test_bvh_strips_equal_full_frame(esc, renderer, bvh_tree):
    """the multi-GPU partition under ESC_STAGE_BVH: 8-row strips sit on the screen bins' 8-row
    grid, every rank's strips reproduce the rows of the full frame"""
    import torch
    from esctp1raytracer_amd import multigpu
    sc, d = synthetic_dict(esc, "c3", 300)
    eye, look = esc.synthetic_view()
    W, H = 200, 93
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    ref = ol.oracle_render(d, eye, look, W, H, threads=8)
    world = 3
    max_rows = multigpu.max_local_rows(H, world)
    gathered = torch.zeros(world, max_rows * W * 3, dtype=torch.float32, device="cuda:0")
    for rank in range(world):
        renderer.render_strips(cam, W, H, rank, world, out_f32=gathered[rank],
                               stage=esc.ESC_STAGE_BVH)
    renderer.synchronize()
    frame = multigpu.assemble_frame_torch(gathered, world, W, H).cpu().numpy()
    assert_bit_equal(frame, ref, "bvh strips")


def test_degenerate_camera_is_refused(esc, renderer):
    """lookfrom == lookat gives a NaN camera basis: every ray would be NaN (the reference renders
    garbage); the render entry points refuse it instead, in every mode"""
    sc, d = synthetic_dict(esc, "c3", 100)
    renderer.upload(sc)
    cam = esc.Camera.for_image((0, 2, -8), (0, 2, -8), 64, 48)
    for stage in (esc.ESC_STAGE_AUTO, esc.ESC_STAGE_LDS, esc.ESC_STAGE_BVH):
        with pytest.raises(esc.EscError):
            renderer.render(cam, 64, 48, stage=stage)
