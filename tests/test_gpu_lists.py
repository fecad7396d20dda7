"""GPU parity of the list-driven sweeps (csrc/rt_lists.h): the tile lists of the primary pass and the
light lists of the shadow pass only ever replace the question "which primitives can this ray touch",
so a frame rendered with them must equal, bit for bit, the frame of the three-level group sweep
(ESC_RENDER_NO_TILE_LISTS | ESC_RENDER_NO_LIGHT_LISTS) and the oracle's.  The cases go after what a
projection can get wrong: primitives behind, around and across the camera plane, cameras inside the
geometry, lists that overflow, bands that do not sit on the tile grid, lights inside the geometry,
shadow rays that start outside the scene (quirk S3), cameras and lights far from the world origin.
The geometry itself is checked per pixel on the CPU in tests/test_tile_lists.py.
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def esc():
    import esctp1raytracer_amd as esc
    return esc


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


def both_ways(esc, renderer, d, eye, look, W, H, what, expect_lists=True, **kw):
    """default (lists) == sweep (lists off) == oracle; returns the list statistics of the default frame"""
    sc = ol.scene_to_product(d)
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    ref = ol.oracle_render(d, eye, look, W, H, threads=8, **{k: v for k, v in kw.items() if k == "shadows"})
    renderer.reset_counters()
    gpu = renderer.render(cam, W, H, **kw)
    c_lists = renderer.counters()
    stats = [renderer.tile_lists(w) for w in (0, 1, 2, 3)]
    assert_bit_equal(gpu, ref, what + "/lists")
    off = esc.ESC_RENDER_NO_TILE_LISTS | esc.ESC_RENDER_NO_LIGHT_LISTS
    renderer.reset_counters()
    sweep = renderer.render(cam, W, H, flags=off, **kw)
    c_sweep = renderer.counters()
    assert_bit_equal(sweep, ref, what + "/sweep")
    for k in ("primary_rays", "hit_pixels", "shadow_rays"):
        assert c_lists[k] == c_sweep[k], (what, k)
    if expect_lists:
        assert any(s is not None for s in stats[:2]), what + ": no tile lists were built"
    return ref, stats


def synthetic_dict(esc, config, n):
    sc = esc.Scene.synthetic(config, n)
    return sc, ol.scene_from_product(sc)


@pytest.mark.parametrize("eye,look", [((0.0, 2.5, -9.0), (3.0, 2.0, -15.0)),    # inside the cloud
                                      ((0.0, 2.5, -9.0), (0.0, 2.5, 5.0)),      # looking back out
                                      ((7.9, 0.6, -2.1), (-8.0, 4.0, -20.0)),   # from a corner
                                      ((0.0, 30.0, -10.0), (0.0, 0.0, -10.01)),  # straight down
                                      ((0.0, 3.0, 40.0), (0.0, 3.0, 80.0)),     # everything behind
                                      ((0.0, 0.7, 3.0), (0.0, 0.7, -30.0))])    # along the floor
def test_sphere_lists_awkward_cameras(esc, renderer, eye, look):
    """spheres behind the camera (their lines cross the image point-mirrored), cut by the camera
    plane (global list), off screen, the camera inside the cloud and grazing the floor"""
    sc, d = synthetic_dict(esc, "c3", 700)
    ref, st = both_ways(esc, renderer, d, eye, look, 232, 140, f"spheres/eye{eye}")
    assert st[0] is not None and st[0]["off"] == 0


@pytest.mark.parametrize("eye,look", [((0.0, 3.0, 6.0), (0.0, 2.0, -8.0)),      # the BASELINE view
                                      ((0.0, 0.45, 3.0), (0.0, 0.3, -30.0)),    # skimming the crests
                                      ((0.0, 0.05, -9.0), (5.0, 0.05, -15.0)),  # inside a valley
                                      ((3.0, 25.0, -10.0), (3.0, 0.0, -10.01)),  # straight down
                                      ((0.0, 3.0, 40.0), (0.0, 3.0, 80.0)),     # everything behind
                                      ((-11.0, 1.0, 3.5), (12.0, 0.0, -24.0))])  # corner to corner
def test_triangle_lists_awkward_cameras(esc, renderer, eye, look):
    """a tessellated heightfield (4,608 triangles): triangles across the camera plane, the camera
    nearly in the planes of many of them (escape bands), grazing views whose tiles hold long lists"""
    sc, d = synthetic_dict(esc, "c5", 48)
    ref, st = both_ways(esc, renderer, d, eye, look, 232, 140, f"heightfield/eye{eye}")
    assert st[1] is not None and st[1]["off"] == 0


def test_camera_in_the_planes_of_a_tessellated_floor(esc, renderer):
    """every floor triangle's plane holds the camera: thousands of cone entries, bands across the
    whole image; beyond kTileEscCap entries the lists switch themselves off and the sweep runs"""
    rng = np.random.default_rng(11)
    for nq, expect_off in ((12, False), (60, True)):
        xs, zs = np.linspace(-6, 6, nq + 1), np.linspace(-14, 2, nq + 1)
        tris = []
        for j in range(nq):
            for i in range(nq):
                a, b = (xs[i], 1.0, zs[j]), (xs[i + 1], 1.0, zs[j])
                c, e = (xs[i + 1], 1.0, zs[j + 1]), (xs[i], 1.0, zs[j + 1])
                tris += [a, b, c, a, c, e]
        geoms = [{"vertex": np.array(tris, np.float32), "face_index": np.arange(len(tris)).reshape(-1, 3),
                  "material": ol.WHITE},
                 {"vertex": np.array([(-0.3, 7, -5), (0.3, 7, -5), (0, 7, -5.6)], np.float32),
                  "face_index": np.array([[0, 1, 2]]), "material": ol.LIGHT_A}]
        sph = np.concatenate([rng.uniform(-4, 4, (90, 1)), rng.uniform(1.2, 3, (90, 1)),
                              rng.uniform(-12, 0, (90, 1)), rng.uniform(0.1, 0.5, (90, 1))], 1)
        mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.2, 0.9, (90, 3))])
        d = ol.scene_dict(geoms, sph.astype(np.float32), mats)
        ref, st = both_ways(esc, renderer, d, (0.0, 1.0, 6.0), (0.0, 1.2, -8.0), 200, 120,
                            f"camera in the floor's plane/{nq}")
        assert st[1] is not None
        assert (st[1]["off"] != 0) == expect_off or st[1]["cones"] > 0
        assert ref.sum() > 0


def test_list_overflow_falls_back_to_the_sweep(esc, renderer):
    """> kTileListCap spheres behind one another in one tile, and > kTileGlobalCap spheres the camera
    plane cuts: those tiles (or the whole frame) take the three-level sweep; same pixels"""
    rng = np.random.default_rng(3)
    n = 700  # > kTileListCap (512)
    line = np.stack([np.full(n, 0.02) + rng.uniform(-0.01, 0.01, n), np.full(n, 1.0),
                     -np.linspace(2.0, 60.0, n), np.linspace(0.05, 0.6, n)], axis=1)
    ang = np.linspace(0.0, 2 * np.pi, 90, endpoint=False)  # a ring in the camera plane z = 3
    around = np.stack([3.0 * np.cos(ang), 1.0 + 3.0 * np.sin(ang), np.full(90, 3.0),
                       np.full(90, 0.4)], axis=1)
    d = ol.load_dump("one")
    for name, sph in (("line", line), ("line + ring", np.concatenate([line, around]))):
        mats = np.stack([ol.material13(ka=c, kd=c) for c in rng.uniform(0.2, 0.9, (len(sph), 3))])
        d2 = ol.scene_dict(d["geometry"], sph.astype(np.float32), mats)
        ref, st = both_ways(esc, renderer, d2, (0, 1, 3), (0, 1, 0), 160, 96, f"overflow/{name}")
        assert st[0] is not None
        if name == "line":
            assert int((st[0]["counts"] > st[0]["cap"]).sum()) > 0, "no tile overflowed"
        else:
            assert st[0]["global"] > st[0]["global_cap"], "the global list did not overflow"
        assert ref.sum() > 0


def _spheres_with_lights(esc, light_tris, n=500):
    sc, d = synthetic_dict(esc, "c3", n)
    geoms = [g for i, g in enumerate(d["geometry"]) if i not in d["light_sources"]]
    for tri in light_tris:
        tri = np.array(tri, np.float32)
        geoms.append({"vertex": tri, "face_index": np.arange(len(tri)).reshape(-1, 3),
                      "material": ol.material13(ka=(.78,) * 3, kd=(.78,) * 3, ke=(17, 12, 4))})
    return ol.scene_dict(geoms, d["spheres"], d["sphere_materials"])


@pytest.mark.parametrize("case", ["light inside the cloud", "light inside a sphere's reach", "two lights",
                                  "five lights", "light far outside", "light level with the floor"])
def test_light_lists_awkward_lights(esc, renderer, case):
    """light lists: a sample point in the middle of the spheres (every face of its cube map is
    busy, spheres cut the faces' planes), a hair outside a sphere but inside its reach (listed for
    every direction), two
    lights (first occluder in index order through the lists; the second light's rays start at
    occluders), five lights (more than get lists), a light far away (all rays in a few cells) and one
    in the floor's plane (rays along the floor)"""
    up = [(-0.5, 12, -9.5), (0.0, 12, -10.5), (0.5, 12, -9.5)]
    mid = [(-0.2, 2.6, -9.8), (0.0, 2.6, -10.2), (0.2, 2.6, -9.8)]
    side = [(-7.5, 6.0, -3.0), (-7.5, 6.5, -3.5), (-7.0, 6.0, -3.0)]
    far = [(300.0, 400.0, 200.0), (300.5, 400.0, 200.0), (300.0, 400.5, 200.0)]
    low = [(9.0, 0.0, 3.0), (9.5, 0.0, 3.0), (9.0, 0.0, 3.5)]
    tris = {"light inside the cloud": [mid], "light inside a sphere's reach": [mid], "two lights": [up, side],
            "five lights": [up, side, mid, [(6, 7, -15), (6, 7.5, -15), (6.5, 7, -15)],
                            [(0, 9, -20), (0.5, 9, -20), (0, 9, -20.5)]],
            "light far outside": [far], "light level with the floor": [low]}[case]
    d = _spheres_with_lights(esc, tris)
    if case == "light inside a sphere":
        d["spheres"][0] = (0.1, 2.6, -9.8, 0.29)  # the sample point (-0.2, 2.6, -9.8) is 0.01 outside it
    eye, look = esc.synthetic_view()
    ref, st = both_ways(esc, renderer, d, eye, look, 224, 128, f"light lists/{case}")
    assert st[2] is not None, "no light lists were built"
    assert ref.sum() > 0


def test_light_lists_fixed_face_of_a_two_face_light(esc, renderer):
    """ESC_FACE_FIXED picks one sample point of a multi-face light for the whole frame: the lists are
    built for that point and rebuilt when the face changes; the hashed choice gets none"""
    sc, d = synthetic_dict(esc, "c3", 400)
    quad = [(-1, 11, -9), (1, 11, -9), (1, 11, -11), (-1, 11, -9), (1, 11, -11), (-1, 11, -11)]
    d = _spheres_with_lights(esc, [quad], 400)
    eye, look = esc.synthetic_view()
    W, H = 200, 112
    renderer.upload(ol.scene_to_product(d))
    cam = esc.Camera.for_image(eye, look, W, H)
    for face in (0, 1, 0):
        ref = ol.oracle_render(d, eye, look, W, H, threads=8, face_mode=0, fixed_face=face)
        gpu = renderer.render(cam, W, H, face_mode=esc.ESC_FACE_FIXED, fixed_face=face)
        assert_bit_equal(gpu, ref, f"fixed face {face}")
        assert renderer.tile_lists(2) is not None
    ref = ol.oracle_render(d, eye, look, W, H, threads=8, face_mode=1, seed=5)
    gpu = renderer.render(cam, W, H, face_mode=esc.ESC_FACE_HASH, seed=5)
    assert_bit_equal(gpu, ref, "hashed faces")


def test_second_light_rays_start_outside_the_scene_box(esc, renderer):
    """quirk S3: light 2's shadow ray starts at camera + dir * (t_occ - eps) with light 1's t2 of the
    occluder -- here a huge sphere far behind the scene, so those origins lie outside the box the
    light lists' reach was computed for and the wave takes the sweep; same pixels"""
    sc, d = synthetic_dict(esc, "c3", 300)
    up = [(-0.5, 12, -9.5), (0.0, 12, -10.5), (0.5, 12, -9.5)]
    side = [(-7.5, 6.0, -3.0), (-7.5, 6.5, -3.5), (-7.0, 6.0, -3.0)]
    d = _spheres_with_lights(esc, [up, side], 300)
    eye, look = esc.synthetic_view()
    ref, st = both_ways(esc, renderer, d, eye, look, 224, 128, "S3 origins")
    assert st[2] is not None


def test_lists_follow_the_camera_and_the_scene(esc, renderer):
    """the tile lists belong to one camera and band, the light lists to one scene: moving the camera,
    resizing the image and uploading another scene must each rebuild what they invalidate"""
    sc, d = synthetic_dict(esc, "c3", 400)
    sc2, d2 = synthetic_dict(esc, "c4", 900)
    views = [((0, 3, 6), (0, 2, -8), 200, 112), ((1, 3, 6), (0, 2, -8), 200, 112),
             ((0, 3, 6), (0, 2, -8), 168, 104), ((0, 3, 6), (0, 2, -8), 200, 112)]
    for scene, dd in ((sc, d), (sc2, d2), (sc, d)):
        renderer.upload(scene)
        for eye, look, W, H in views:
            cam = esc.Camera.for_image(eye, look, W, H)
            gpu = renderer.render(cam, W, H)
            ref = ol.oracle_render(dd, eye, look, W, H, threads=8)
            assert_bit_equal(gpu, ref, f"moving/{eye}/{W}x{H}")


def test_lists_on_bands_and_strips(esc, renderer):
    """a band that does not start on a multiple of 4 rows cannot use the tile lists (sweep), aligned
    bands and the multi-GPU strips can; every piece reproduces its rows of the full frame"""
    import torch
    from esctp1raytracer_amd import multigpu
    sc, d = synthetic_dict(esc, "c3", 500)
    eye, look = esc.synthetic_view()
    W, H = 200, 93
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    ref = ol.oracle_render(d, eye, look, W, H, threads=8)
    for r0, r1 in ((13, 53), (12, 52), (0, 93), (88, 93), (4, 5)):
        band = torch.zeros((r1 - r0) * W * 3, dtype=torch.float32, device="cuda:0")
        renderer.render_rows(cam, W, H, r0, r1, out_f32=band)
        renderer.synchronize()
        assert_bit_equal(band.cpu().numpy().reshape(r1 - r0, W, 3), ref[r0:r1], f"band {r0}:{r1}")
        assert (renderer.tile_lists(0) is not None) == (r0 % 4 == 0), f"band {r0}:{r1}"
    for world in (2, 3):
        max_rows = multigpu.max_local_rows(H, world)
        gathered = torch.zeros(world, max_rows * W * 3, dtype=torch.float32, device="cuda:0")
        for rank in range(world):
            renderer.render_strips(cam, W, H, rank, world, out_f32=gathered[rank])
            assert renderer.tile_lists(0) is not None
        renderer.synchronize()
        frame = multigpu.assemble_frame_torch(gathered, world, W, H).cpu().numpy()
        assert_bit_equal(frame, ref, f"strips/world {world}")


def test_lists_far_from_the_world_origin(esc, renderer):
    """camera, scene and light 2,000 units from the origin: the rays' fp32 rounding is dozens of
    pixels' worth of image-plane coordinates there and the rectangles must grow with it"""
    rng = np.random.default_rng(21)
    off = np.array([1500.0, -700.0, 900.0])
    n = 400
    c = np.concatenate([rng.uniform(-4, 4, (n, 1)), rng.uniform(0.3, 3.5, (n, 1)),
                        rng.uniform(-12, -1, (n, 1))], axis=1) + off
    r = rng.uniform(0.05, 0.4, n)
    fl = np.array([[-8, 0, 4], [8, 0, 4], [8, 0, -14], [-8, 0, -14]], float) + off
    l1 = np.array([[-0.3, 9, -5], [0.3, 9, -5], [0, 9, -5.6]], float) + off
    geoms = [{"vertex": fl[[0, 1, 2, 0, 2, 3]].astype(np.float32), "face_index": np.arange(6).reshape(2, 3),
              "material": ol.WHITE},
             {"vertex": l1.astype(np.float32), "face_index": np.array([[0, 1, 2]]), "material": ol.LIGHT_A}]
    sph = np.concatenate([c, r[:, None]], axis=1).astype(np.float32)
    mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.2, 0.9, (n, 3))])
    d = ol.scene_dict(geoms, sph, mats)
    eye, look = tuple(np.array([0.0, 2.0, 6.0]) + off), tuple(np.array([0.0, 1.5, -6.0]) + off)
    ref, st = both_ways(esc, renderer, d, eye, look, 224, 128, "far from the origin")
    assert ref.sum() > 0


@pytest.mark.parametrize("seed", list(range(12)))
def test_lists_random_scenes(esc, renderer, seed):
    """random triangle soup (grouped: >= 64) + spheres + 1..3 one-face lights, random camera inside
    or outside: lists == sweep == oracle"""
    rng = np.random.default_rng(5000 + seed)
    n_tri = int(rng.integers(64, 900))
    n_sph = int(rng.integers(64, 600))
    c = rng.uniform(-4, 4, (n_tri, 1, 3))
    tri = (c + rng.normal(0, rng.uniform(0.05, 1.5), (n_tri, 3, 3))).astype(np.float32)
    if seed % 3 == 0:  # slivers and points among them
        tri[::17, 2] = tri[::17, 1] + (tri[::17, 1] - tri[::17, 0]) * 1e-6
        tri[::29, 1] = tri[::29, 0]
    geoms = []
    per = max(1, n_tri // 5)
    for k in range(0, n_tri, per):
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
    mats = np.stack([ol.material13(ka=m, kd=m) for m in rng.uniform(0.1, 0.9, (n_sph, 3))])
    d = ol.scene_dict(geoms, sph.astype(np.float32), mats)
    eye = tuple(float(x) for x in rng.uniform(-6, 6, 3))
    look = tuple(float(x) for x in rng.uniform(-2, 2, 3))
    both_ways(esc, renderer, d, eye, look, 168, 104, f"random/{seed}")


@pytest.mark.parametrize("config,W,H", [("c3", 3840, 2160), ("c4", 3840, 2160), ("c5", 7680, 4320)])
def test_lists_equal_sweep_full_size(esc, renderer, config, W, H):
    """the BASELINE frames at their own size: every fp32 value, quantised byte and ray counter of the
    list-driven frame equals the three-level sweep's (which test_gpu_parity compares with the oracle)"""
    import torch
    sc = esc.Scene.synthetic(config)
    eye, look = esc.synthetic_view()
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    a = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
    b = torch.zeros_like(a)
    a8 = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda:0")
    b8 = torch.zeros_like(a8)
    renderer.reset_counters()
    renderer.render_rows(cam, W, H, 0, H, out_f32=a, out_u8=a8)
    ca = renderer.counters()
    st = [renderer.tile_lists(w) for w in (0, 1, 2, 3)]
    renderer.reset_counters()
    renderer.render_rows(cam, W, H, 0, H, out_f32=b, out_u8=b8,
                         flags=esc.ESC_RENDER_NO_TILE_LISTS | esc.ESC_RENDER_NO_LIGHT_LISTS)
    cb = renderer.counters()
    renderer.synchronize()
    assert int((a.view(torch.int32) != b.view(torch.int32)).sum().item()) == 0
    assert bool(torch.equal(a8, b8))
    for k in ("primary_rays", "hit_pixels", "shadow_rays"):
        assert ca[k] == cb[k]
    which = 1 if config == "c5" else 0
    assert st[which] is not None and st[which]["off"] == 0
    frac_over = float((st[which]["counts"] > st[which]["cap"]).mean())
    assert frac_over < 0.2, f"{frac_over:.2%} of the tiles overflow"
    if config != "c5":
        assert st[2] is not None


@pytest.mark.parametrize("stage", ["auto", "bvh"])
def test_recorded_frames_replay_the_same_frame(esc, renderer, stage):
    """esc_frame_record / esc_frame_launch: one frame's launches as a HIP graph.  Replays write the
    frame the plain entry writes (and count the same rays); once the context has rendered another
    camera the recorded frame is refused (its kernels would read rebuilt per-camera tables), and a
    new recording works again.  Rank 1 of 3 as well as the whole frame."""
    import torch
    st = {"auto": esc.ESC_STAGE_AUTO, "bvh": esc.ESC_STAGE_BVH}[stage]
    sc, d = synthetic_dict(esc, "c3", 500)
    eye, look = esc.synthetic_view()
    W, H = 200, 96
    renderer.upload(sc)
    cam = esc.Camera.for_image(eye, look, W, H)
    ref = ol.oracle_render(d, eye, look, W, H, threads=8)
    for first, stride in ((0, 1), (1, 3)):
        rows = esc.strip_local_rows(H, 8, first, stride)
        plain = torch.zeros(rows * W * 3, dtype=torch.float32, device="cuda:0")
        renderer.render_strips(cam, W, H, first, stride, out_f32=plain, stage=st)
        renderer.synchronize()
        if stride == 1:
            assert_bit_equal(plain.cpu().numpy().reshape(H, W, 3), ref, "plain frame")
        out = torch.zeros_like(plain)
        u8 = torch.zeros(rows * W * 3, dtype=torch.uint8, device="cuda:0")
        rec = renderer.record_strips(cam, W, H, first, stride, out_f32=out, out_u8=u8, stage=st)
        renderer.reset_counters()
        renderer.render_strips(cam, W, H, first, stride, out_f32=plain, stage=st)
        one = renderer.counters()
        renderer.reset_counters()
        for _ in range(3):
            out.zero_()
            rec.launch()
            renderer.synchronize()
            assert bool(torch.equal(out.view(torch.int32), plain.view(torch.int32)))
        three = renderer.counters()
        for k in ("primary_rays", "hit_pixels", "shadow_rays"):
            assert three[k] == 3 * one[k]
        assert np.array_equal(u8.cpu().numpy(), ol.oracle_quantise(plain.cpu().numpy()))
        # another camera on the same context: the recording is stale and says so
        renderer.render_strips(esc.Camera.for_image((1, 3, 6), look, W, H), W, H, first, stride,
                               out_f32=plain, stage=st)
        with pytest.raises(esc.EscError):
            rec.launch()
        rec.close()
        rec = renderer.record_strips(cam, W, H, first, stride, out_f32=out, stage=st)
        out.zero_()
        rec.launch()
        renderer.synchronize()
        renderer.render_strips(cam, W, H, first, stride, out_f32=plain, stage=st)
        renderer.synchronize()
        assert bool(torch.equal(out.view(torch.int32), plain.view(torch.int32)))
        rec.close()


def test_random_scenes_lists_equal_sweep_and_oracle(esc, renderer):
    """tests/random_scenes.py: sphere clouds, heightfield patches, loose triangles and slivers, one to three
    lights anywhere, cameras anywhere, odd image sizes -- the default frame (lists) == the group sweep
    (lists off) on every scene, == the index-order sweep on every fourth, == the oracle on every
    tenth.  (tools/list_hunt.py runs the same generator for as long as one likes: 171,000 scenes on
    the final lists of round 3, 21,000 of them also against the oracle, 0 differences.)"""
    from random_scenes import random_scene
    off = esc.ESC_RENDER_NO_TILE_LISTS | esc.ESC_RENDER_NO_LIGHT_LISTS
    lit = listed = 0
    for seed in range(200000, 200080):
        d, eye, look, W, H, vfov = random_scene(seed)
        renderer.upload(ol.scene_to_product(d))
        cam = esc.Camera.for_image(eye, look, W, H, vfov=vfov)
        a = renderer.render(cam, W, H)
        st = [renderer.tile_lists(w) for w in range(4)]
        b = renderer.render(cam, W, H, flags=off)
        assert_bit_equal(a, b, f"seed {seed}: lists vs sweep")
        if seed % 4 == 0:
            c = renderer.render(cam, W, H, flags=esc.ESC_RENDER_INDEX_ORDER)
            assert_bit_equal(a, c, f"seed {seed}: lists vs index order")
        if seed % 10 == 0:
            ref = ol.oracle_render(d, eye, look, W, H, threads=8, vfov=vfov)
            assert_bit_equal(a, ref, f"seed {seed}: lists vs oracle")
        lit += 1 if a.any() else 0
        listed += 1 if any(s is not None and s["off"] == 0 for s in st) else 0
    assert lit > 30 and listed > 50, (lit, listed)


def test_no_counters_flag_renders_the_same_frame_and_counts_nothing(esc, renderer):
    """ESC_RENDER_NO_COUNTERS: the instrumentation is off for that call -- same image bit for bit,
    the counters stay where they were (bench.py times such frames and counts the rays on one other)"""
    sc, d = synthetic_dict(esc, "c3", 600)
    renderer.upload(sc)
    eye, look = esc.synthetic_view()
    cam = esc.Camera.for_image(eye, look, 200, 120)
    for flags in (0, esc.ESC_RENDER_TWO_KERNELS, esc.ESC_RENDER_INDEX_ORDER):
        renderer.reset_counters()
        a = renderer.render(cam, 200, 120, flags=flags)
        c1 = renderer.counters()
        b = renderer.render(cam, 200, 120, flags=flags | esc.ESC_RENDER_NO_COUNTERS)
        c2 = renderer.counters()
        assert_bit_equal(b, a, f"flags {flags}: without counters")
        assert c1 == c2 and c1["primary_rays"] == 200 * 120, (c1, c2)
    buf = np.full((120, 200, 3), 7.0, np.float32)  # a host array the caller owns and reuses
    assert renderer.render(cam, 200, 120, out=buf) is buf
    assert_bit_equal(buf, a, "render(out=...)")
