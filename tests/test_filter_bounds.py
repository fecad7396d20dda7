"""The conservative sphere FILTERS of the brute-force kernels (csrc/rt_brute.h "FILTERS"), checked
on the CPU: a numpy mirror of the filter expressions against the reference's own fp32
arithmetic (restated here operation by operation) on millions of adversarial
(ray, sphere) pairs that graze the silhouette.

Property (what the kernels rely on):   the reference does not reject at `disc < 0`  ==>  q' >= 0.
The proof is in rt_brute.h; this is the experiment that would catch a slip in it.  The GPU side
of the same claim is tests/test_gpu_parity.py::test_filter_equals_exact_only (whole frames, the
filtered kernels against ESC_RENDER_EXACT_ONLY and the oracle).

fp32 fma is emulated as float32(float64(a) * float64(b) + float64(c)): the product is exact in
double; the double rounding of the sum can differ from a true fma by one fp32 ulp at most, and
never in sign -- the margins (32u / 256u against 13.3u / ~100u needed) dwarf it.
"""
import numpy as np
import pytest

f32 = np.float32
U = 2.0 ** -24


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def ref_dot(ax, ay, az, bx, by, bz):  # vec.h:95-101 order, no fusion
    return f32(f32(f32(ax * bx) + f32(ay * by)) + f32(az * bz))


def unit(v):
    n = np.sqrt((v.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    return (v / n).astype(f32)


def grazing_rays(rng, o, c, r, n):
    """n unit directions from o that pass the sphere (c, r) at (1 + delta) r, |delta| from 1e-7
    to 1e-2, both signs: half would hit, half would miss, all by a hair."""
    oc = (c - o).astype(np.float64)
    dist = np.linalg.norm(oc, axis=1, keepdims=True)
    w = oc / dist
    a = rng.normal(size=(n, 3))
    perp = a - (a * w).sum(axis=1, keepdims=True) * w
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    delta = rng.choice([-1.0, 1.0], size=(n, 1)) * 10.0 ** rng.uniform(-7, -2, size=(n, 1))
    s = np.clip(r.astype(np.float64)[:, None] * (1.0 + delta) / dist, 0.0, 0.999999)
    d = w * np.sqrt(1.0 - s * s) + perp * s
    return unit(d)


@pytest.mark.parametrize("scale", [1.0, 30.0, 1000.0])
def test_primary_filter_never_rejects_a_reference_candidate(scale):
    """k_prepare_primary's DevSphF + sph4_primary_filter_pk vs the hoisted reference test
    (b = dot(oc, d); disc = b*b - cc)."""
    rng = np.random.default_rng(int(scale))
    n = 1_500_000
    o = (rng.uniform(-1, 1, (n, 3)) * scale).astype(f32)
    c = (rng.uniform(-1, 1, (n, 3)) * scale).astype(f32)
    r = (10.0 ** rng.uniform(-3, 0, n) * scale * 0.2).astype(f32)
    d = grazing_rays(rng, o, c, r, n)
    # hoisted per-sphere values exactly as k_prepare_primary computes them
    ocx, ocy, ocz = f32(o[:, 0] - c[:, 0]), f32(o[:, 1] - c[:, 1]), f32(o[:, 2] - c[:, 2])
    r2 = f32(r * r)
    cc = f32(ref_dot(ocx, ocy, ocz, ocx, ocy, ocz) - r2)
    A = f32(f32(np.abs(ocx) + np.abs(ocy)) + np.abs(ocz))
    ccm = f32(cc - f32(f32(f32(f32(A * A) + np.abs(r2)) * f32(2.0 ** -19)) + f32(2.0 ** -120)))
    # reference: not rejected at `disc < 0`
    b = ref_dot(ocx, ocy, ocz, d[:, 0], d[:, 1], d[:, 2])
    disc = f32(f32(b * b) - cc)
    ref_candidate = ~(disc < 0)
    # filter, unscaled form of the proof: b'^2 >= ccm
    bf = fma(ocz, d[:, 2], fma(ocy, d[:, 1], f32(ocx * d[:, 0])))
    qf = fma(bf, bf, -ccm)
    assert not (ref_candidate & ~(qf >= 0)).any()
    # filter as the kernel runs it (k_prepare_primary's DevSphF + sph4_primary_filter_pk):
    # oc scaled by 1/s, candidate iff |b''| >= 1; spheres with ccm <= 0 or s <= 0: always
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(np.maximum(ccm, f32(0)), dtype=f32)
        s_eff = f32(f32(f32(sq * f32(1 - 2.0 ** -22)) - f32(A * f32(9 * 2.0 ** -24))) * f32(1 - 2.0 ** -22))
        ok = (ccm > 0) & (s_eff > 0)
        inv = f32(f32(1) / np.where(ok, s_eff, f32(1)))
    sx = np.where(ok, f32(ocx * inv), f32(0))
    sy = np.where(ok, f32(ocy * inv), f32(0))
    sz = np.where(ok, f32(ocz * inv), f32(0))
    w = np.where(ok, f32(0), f32(2))

    def scaled(dd):
        return np.abs(fma(sz, dd[:, 2], fma(sy, dd[:, 1], fma(sx, dd[:, 0], w)))) >= 1

    filt_candidate = scaled(d)
    assert ref_candidate.sum() > n // 4 and (~ref_candidate).sum() > n // 4
    missed = ref_candidate & ~filt_candidate
    assert not missed.any(), f"{int(missed.sum())} reference candidates rejected by the filter"
    # and it is a FILTER: away from the silhouette it rejects (here every ray grazes within 1 %,
    # so most are passed; a random direction must not be)
    assert scaled(unit(rng.normal(size=(n, 3)))).mean() < 0.25


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (30.0, 0.0), (30.0, 500.0), (1000.0, 0.0)])
def test_shadow_filter_never_rejects_a_reference_candidate(scale, offset):
    """commit()'s DevSphPairF + make_ray_filter + pair4_any_filter_pk vs the reference's 16-op
    shadow test (oc = o - c; b = dot(oc, L); cc = dot(oc, oc) - r2; disc = b*b - cc).  `offset`
    moves the whole scene away from the world origin: the filter works relative to the scene
    point g, so its margins must not care."""
    rng = np.random.default_rng(7 + int(scale) + int(offset))
    n = 1_500_000
    o = (rng.uniform(-1, 1, (n, 3)) * scale + offset).astype(f32)
    c = (rng.uniform(-1, 1, (n, 3)) * scale + offset).astype(f32)
    r = (10.0 ** rng.uniform(-3, 0, n) * scale * 0.2).astype(f32)
    L = grazing_rays(rng, o, c, r, n)
    r2 = f32(r * r)
    # reference
    ocx, ocy, ocz = f32(o[:, 0] - c[:, 0]), f32(o[:, 1] - c[:, 1]), f32(o[:, 2] - c[:, 2])
    b = ref_dot(ocx, ocy, ocz, L[:, 0], L[:, 1], L[:, 2])
    cc = f32(ref_dot(ocx, ocy, ocz, ocx, ocy, ocz) - r2)
    disc = f32(f32(b * b) - cc)
    ref_candidate = ~(disc < 0)
    # host side of the filter (rt_capi.cpp commit()): g, c' = fl(c - g), km rounded up from double
    g = (0.5 * (c.min(axis=0).astype(np.float64) + c.max(axis=0).astype(np.float64))).astype(f32)
    cp = (c.astype(np.float64) - g.astype(np.float64)).astype(f32)
    c2 = (cp.astype(np.float64) ** 2).sum(axis=1)
    km_d = r2.astype(np.float64) - c2 + 2.0 ** -16 * (c2 + np.abs(r2.astype(np.float64))) + 2.0 ** -120
    km = km_d.astype(f32)
    low = km.astype(np.float64) < km_d
    km[low] = np.nextafter(km[low], f32(np.inf))
    # device side (make_ray_filter, un-fused) ...
    ax, ay, az = f32(o[:, 0] - g[0]), f32(o[:, 1] - g[1]), f32(o[:, 2] - g[2])
    nn = ref_dot(ax, ay, az, ax, ay, az)
    ss = ref_dot(ax, ay, az, L[:, 0], L[:, 1], L[:, 2])
    nko = f32(nn * f32(-(1.0 - 2.0 ** -16)))
    ms = f32(-ss)
    # ... and pair4_any_filter_pk
    y = fma(cp[:, 2], f32(az + az), fma(cp[:, 1], f32(ay + ay), fma(cp[:, 0], f32(ax + ax), nko)))
    x = fma(cp[:, 2], L[:, 2], fma(cp[:, 1], L[:, 1], fma(cp[:, 0], L[:, 0], ms)))
    q = f32(fma(x, x, y) + km)
    filt_candidate = q >= 0
    assert ref_candidate.sum() > n // 4 and (~ref_candidate).sum() > n // 4
    missed = ref_candidate & ~filt_candidate
    assert not missed.any(), f"{int(missed.sum())} reference candidates rejected by the filter"
    Lr = unit(rng.normal(size=(n, 3)))
    ssr = ref_dot(ax, ay, az, Lr[:, 0], Lr[:, 1], Lr[:, 2])
    xr = fma(cp[:, 2], Lr[:, 2], fma(cp[:, 1], Lr[:, 1], fma(cp[:, 0], Lr[:, 0], f32(-ssr))))
    assert (f32(fma(xr, xr, y) + km) >= 0).mean() < 0.25


def _scaled_record(ocx, ocy, ocz, cc, r2a):
    """k_prepare_*'s sphere_filter_record: (sx, sy, sz, w) of the scaled test |b''| >= 1"""
    A = f32(f32(np.abs(ocx) + np.abs(ocy)) + np.abs(ocz))
    ccm = f32(cc - f32(f32(f32(f32(A * A) + r2a) * f32(2.0 ** -19)) + f32(2.0 ** -120)))
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(np.maximum(ccm, f32(0)), dtype=f32)
        s_eff = f32(f32(f32(sq * f32(1 - 2.0 ** -22)) - f32(A * f32(9 * 2.0 ** -24))) * f32(1 - 2.0 ** -22))
        ok = (ccm > 0) & (s_eff > 0)
        inv = f32(f32(1) / np.where(ok, s_eff, f32(1)))
    z = f32(0)
    return (np.where(ok, f32(ocx * inv), z), np.where(ok, f32(ocy * inv), z),
            np.where(ok, f32(ocz * inv), z), np.where(ok, z, f32(2)))


def _groups(rng, n_grp, scale, spread):
    """n_grp groups of 8 spheres: centres within `spread` of a group point, radii 1e-3..1 x 0.2
    scale; returns member centres/radii [n_grp, 8], and group_bounds()'s (C, rgeo) in fp32"""
    gc = rng.uniform(-1, 1, (n_grp, 1, 3)) * scale
    c = (gc + rng.normal(0, 1, (n_grp, 8, 3)) * spread).astype(f32)
    r = (10.0 ** rng.uniform(-3, 0, (n_grp, 8)) * scale * 0.2).astype(f32)
    r2 = f32(r * r)
    rd = np.sqrt(r2.astype(np.float64))
    lo = (c.astype(np.float64) - rd[..., None]).min(axis=1)
    hi = (c.astype(np.float64) + rd[..., None]).max(axis=1)
    C = (0.5 * (lo + hi)).astype(f32)
    dist = np.linalg.norm(c.astype(np.float64) - C.astype(np.float64)[:, None, :], axis=2)
    rg_d = (rd + dist).max(axis=1) * (1.0 + 2.0 ** -40)
    rg = rg_d.astype(f32)
    low = rg.astype(np.float64) < rg_d
    rg[low] = np.nextafter(rg[low], f32(np.inf))
    return c, r2, C, rg


@pytest.mark.parametrize("scale,spread", [(1.0, 0.05), (30.0, 1.0), (1000.0, 5.0), (1000.0, 0.01)])
def test_primary_group_filter_never_rejects_a_member_candidate(scale, spread):
    """rt_device.h SphGroups, primary rays: k_prepare_groups' bounding-sphere record (R = rgeo +
    0x1.2p-10 (A_G + 2 rgeo) + 2^-60) against the reference test of every member: a ray for which
    ANY member is not rejected at `disc < 0` must have |b''_G| >= 1.  Rays graze a member's
    silhouette; members reach from well inside the group to its rim, specks included."""
    rng = np.random.default_rng(int(scale * 10 + spread * 1000))
    n_grp = 200_000
    c, r2, C, rg = _groups(rng, n_grp, scale, spread * scale / 30.0 if scale > 1 else spread)
    o = (rng.uniform(-1, 1, (n_grp, 3)) * scale * 3).astype(f32)
    k = rng.integers(0, 8, n_grp)
    idx = np.arange(n_grp)
    d = grazing_rays(rng, o, c[idx, k], np.sqrt(r2[idx, k]), n_grp)
    # reference, every member (hoisted as k_prepare_groups does)
    any_cand = np.zeros(n_grp, bool)
    for m in range(8):
        ocx, ocy, ocz = f32(o[:, 0] - c[:, m, 0]), f32(o[:, 1] - c[:, m, 1]), f32(o[:, 2] - c[:, m, 2])
        cc = f32(ref_dot(ocx, ocy, ocz, ocx, ocy, ocz) - r2[:, m])
        b = ref_dot(ocx, ocy, ocz, d[:, 0], d[:, 1], d[:, 2])
        any_cand |= ~(f32(f32(b * b) - cc) < 0)
    # group record
    gx, gy, gz = f32(o[:, 0] - C[:, 0]), f32(o[:, 1] - C[:, 1]), f32(o[:, 2] - C[:, 2])
    A = f32(f32(np.abs(gx) + np.abs(gy)) + np.abs(gz))
    R = f32(f32(rg + f32(f32(1.125 * 2.0 ** -10) * f32(A + f32(f32(2) * rg)))) + f32(2.0 ** -60))
    R2 = f32(f32(R * R) * f32(1.00001))
    sx, sy, sz, w = _scaled_record(gx, gy, gz, f32(ref_dot(gx, gy, gz, gx, gy, gz) - R2), R2)
    passed = np.abs(fma(sz, d[:, 2], fma(sy, d[:, 1], fma(sx, d[:, 0], w)))) >= 1
    assert any_cand.sum() > n_grp // 4
    missed = any_cand & ~passed
    assert not missed.any(), f"{int(missed.sum())} member candidates behind a rejected group"
    dr = unit(rng.normal(size=(n_grp, 3)))
    assert (np.abs(fma(sz, dr[:, 2], fma(sy, dr[:, 1], fma(sx, dr[:, 0], w)))) >= 1).mean() < 0.5


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (30.0, 0.0), (30.0, 500.0), (1000.0, 0.0)])
def test_shadow_group_filter_never_rejects_a_member_candidate(scale, offset):
    """rt_device.h SphGroups, shadow rays of the last light: commit()'s bounding-sphere record in
    DevSphPairF form (R = rgeo + 0x1.6p-10 (rho_max + |C - g| + rgeo) + 2^-60) against the reference's
    16-op test of every member, for origins inside the scene box (|fl(O - g)|_1 <= rho_max)."""
    rng = np.random.default_rng(11 + int(scale) + int(offset))
    n_grp = 200_000
    c, r2, C, rg = _groups(rng, n_grp, scale, 0.03 * scale)
    c = (c.astype(np.float64) + offset).astype(f32)
    C = (C.astype(np.float64) + offset).astype(f32)  # bounds stay valid up to fp32 rounding of
    rg = f32(rg * f32(1.0 + 2.0 ** -20) + np.abs(C).max() * f32(2.0 ** -22))  # the shifted centres
    o = (rng.uniform(-1, 1, (n_grp, 3)) * scale + offset).astype(f32)
    k = rng.integers(0, 8, n_grp)
    idx = np.arange(n_grp)
    L = grazing_rays(rng, o, c[idx, k], np.sqrt(r2[idx, k]), n_grp)
    any_cand = np.zeros(n_grp, bool)
    for m in range(8):
        ocx, ocy, ocz = f32(o[:, 0] - c[:, m, 0]), f32(o[:, 1] - c[:, m, 1]), f32(o[:, 2] - c[:, m, 2])
        b = ref_dot(ocx, ocy, ocz, L[:, 0], L[:, 1], L[:, 2])
        cc = f32(ref_dot(ocx, ocy, ocz, ocx, ocy, ocz) - r2[:, m])
        any_cand |= ~(f32(f32(b * b) - cc) < 0)
    # host side: g and rho_max as commit() takes them (box of everything, 1-norm radius doubled)
    rd = np.sqrt(r2.astype(np.float64))
    lo = np.minimum((c.astype(np.float64) - rd[..., None]).min(axis=(0, 1)), o.min(axis=0))
    hi = np.maximum((c.astype(np.float64) + rd[..., None]).max(axis=(0, 1)), o.max(axis=0))
    g = (0.5 * (lo + hi)).astype(f32)
    rho = 2.0 * np.maximum(hi - g, g - lo).sum() + 1e-30
    Cg = C.astype(np.float64) - g.astype(np.float64)
    Rd = rg.astype(np.float64) + 1.375 * 2.0 ** -10 * (rho + np.linalg.norm(Cg, axis=1) + rg) + 2.0 ** -60
    R2 = Rd * Rd * 1.00001
    cp = Cg.astype(f32)
    c2 = (cp.astype(np.float64) ** 2).sum(axis=1)
    km_d = R2 - c2 + 2.0 ** -16 * (c2 + R2) + 2.0 ** -120
    km = km_d.astype(f32)
    low = km.astype(np.float64) < km_d
    km[low] = np.nextafter(km[low], f32(np.inf))
    ax, ay, az = f32(o[:, 0] - g[0]), f32(o[:, 1] - g[1]), f32(o[:, 2] - g[2])
    assert (f32(f32(np.abs(ax) + np.abs(ay)) + np.abs(az)) <= f32(rho)).all()  # none is `far`
    nn = ref_dot(ax, ay, az, ax, ay, az)
    ss = ref_dot(ax, ay, az, L[:, 0], L[:, 1], L[:, 2])
    nko = f32(nn * f32(-(1.0 - 2.0 ** -16)))
    y = fma(cp[:, 2], f32(az + az), fma(cp[:, 1], f32(ay + ay), fma(cp[:, 0], f32(ax + ax), nko)))
    x = fma(cp[:, 2], L[:, 2], fma(cp[:, 1], L[:, 1], fma(cp[:, 0], L[:, 0], f32(-ss))))
    passed = f32(fma(x, x, y) + km) >= 0
    assert any_cand.sum() > n_grp // 4
    missed = any_cand & ~passed
    assert not missed.any(), f"{int(missed.sum())} member candidates behind a rejected group"


def test_group_slack_constants():
    """the constants the proofs in rt_brute.h quote for the group radii"""
    assert 1.125 * 2.0 ** -10 > (19.4 ** 0.5) * 2.0 ** -12 + 2.1 * U   # primary: 0x1.2p-10
    assert 1.375 * 2.0 ** -10 > (29.1 ** 0.5) * 2.0 ** -12               # shadow:  0x1.6p-10
    assert float.fromhex("0x1.2p-10") == 1.125 * 2.0 ** -10
    assert float.fromhex("0x1.6p-10") == 1.375 * 2.0 ** -10


@pytest.mark.parametrize("slack,expect_miss", [(1.125 * 2.0 ** -10, False), (0.0, True)])
def test_primary_group_slack_is_what_specks_need(slack, expect_miss):
    """Specks (r << sqrt(u) x distance): the reference's `disc < 0` is decided by rounding noise
    and lets rays through that pass the speck at hundreds of radii.  A speck on the rim of its
    group, the ray passing on the far side: the group radius must reach rgeo + that noise.  With
    the documented slack no member candidate is lost, without it some are -- the experiment has
    the power to see a missing term."""
    rng = np.random.default_rng(99)
    n = 400_000
    C = (rng.uniform(-1, 1, (n, 3)) * 100).astype(np.float64)
    o = (C + unit(rng.normal(size=(n, 3))) * 10.0 ** rng.uniform(1.5, 3, (n, 1))).astype(f32)
    dist = np.linalg.norm(o.astype(np.float64) - C, axis=1, keepdims=True)
    w = (C - o) / dist
    e = rng.normal(size=(n, 3))
    e -= (e * w).sum(axis=1, keepdims=True) * w
    e /= np.linalg.norm(e, axis=1, keepdims=True)            # across the line of sight
    half = dist * 10.0 ** rng.uniform(-3.0, -1.5, (n, 1))     # group half-width: the noise reach and up
    c1 = (C + e * half).astype(f32)                           # the speck on the rim ...
    c2 = (C - e * half).astype(f32)                           # ... and its opposite number
    r2 = f32(1e-10)
    Cf = (0.5 * (np.minimum(c1, c2).astype(np.float64) - 1e-5 + np.maximum(c1, c2).astype(np.float64) + 1e-5)).astype(f32)
    rg_d = np.maximum(np.linalg.norm(c1.astype(np.float64) - Cf, axis=1),
                      np.linalg.norm(c2.astype(np.float64) - Cf, axis=1)) + 1e-5
    rg = (rg_d * (1 + 2.0 ** -40)).astype(f32)
    rg = np.where(rg.astype(np.float64) < rg_d, np.nextafter(rg, f32(np.inf)), rg)
    D = dist * rng.uniform(0, 8e-4, (n, 1))                   # beyond the speck, outward
    d = unit(c1.astype(np.float64) + e * D - o)
    ocx, ocy, ocz = f32(o[:, 0] - c1[:, 0]), f32(o[:, 1] - c1[:, 1]), f32(o[:, 2] - c1[:, 2])
    cc = f32(ref_dot(ocx, ocy, ocz, ocx, ocy, ocz) - r2)
    b = ref_dot(ocx, ocy, ocz, d[:, 0], d[:, 1], d[:, 2])
    cand = ~(f32(f32(b * b) - cc) < 0)
    gx, gy, gz = f32(o[:, 0] - Cf[:, 0]), f32(o[:, 1] - Cf[:, 1]), f32(o[:, 2] - Cf[:, 2])
    A = f32(f32(np.abs(gx) + np.abs(gy)) + np.abs(gz))
    R = f32(f32(rg + f32(f32(slack) * f32(A + f32(f32(2) * rg)))) + f32(2.0 ** -60))
    R2 = f32(f32(R * R) * f32(1.00001))
    sx, sy, sz, w4 = _scaled_record(gx, gy, gz, f32(ref_dot(gx, gy, gz, gx, gy, gz) - R2), R2)
    passed = np.abs(fma(sz, d[:, 2], fma(sy, d[:, 1], fma(sx, d[:, 0], w4)))) >= 1
    assert cand.sum() > n // 20 and (~cand).sum() > n // 20
    assert bool((cand & ~passed).any()) == expect_miss


def test_filter_margins_match_the_documented_budget():
    """the constants the proof in rt_brute.h quotes: provided margins exceed the needed ones"""
    assert 32 * U == 2.0 ** -19 and 256 * U == 2.0 ** -16
    assert 32 > 13.3 + 1 + 5 * 32 * U          # primary: 13.3u + rounding of ccm + of A2f
    assert 251 > 99.4 + 4.01 and 255 > 99.4     # shadow: ray side (nko), sphere side (km)


# ------------------------------------------------------------------ triangles
def ref_cross(ax, ay, az, bx, by, bz):  # vec.h:103, no fusion
    return (f32(f32(ay * bz) - f32(az * by)), f32(f32(az * bx) - f32(ax * bz)),
            f32(f32(ax * by) - f32(ay * bx)))


def l1(x, y, z):
    return f32(f32(np.abs(x) + np.abs(y)) + np.abs(z))


def fdot(ax, ay, az, bx, by, bz):  # mul, fma, fma
    return fma(az, bz, fma(ay, by, f32(ax * bx)))


def uv_accept(det, un, vn):
    """ray_triangle.h:21-41 on the fp32 numerators: every reject that involves det, u, v (the t
    rejects only shrink the accepted set, so passing these is implied by an accept)"""
    eps = np.float64(np.finfo(np.float32).eps)
    d = det.astype(np.float64)
    ok = ~((d > -eps) & (d < eps))
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        u2 = (un.astype(np.float64) * inv).astype(f32)
        v2 = (vn.astype(np.float64) * inv).astype(f32)
    ok &= ~((u2 < f32(eps)) | (u2 > f32(1)))
    ok &= ~((v2 < f32(eps)) | (f32(u2 + v2) > f32(1)))
    return ok


def rays_near_edges(rng, o, v0, e1, e2, n):
    """directions from o to points of the triangle's plane whose barycentrics hug the boundary
    (u, v or 1-u-v within 1e-7..1e-2 of 0, either side), so hits and misses are a hair apart"""
    u = rng.uniform(0, 1, n)
    v = rng.uniform(0, 1, n) * (1 - u)
    which = rng.integers(0, 3, n)
    delta = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-7, -2, n)
    u = np.where(which == 0, delta, u)
    v = np.where(which == 1, delta, v)
    v = np.where(which == 2, 1 - u + delta, v)
    p = v0.astype(np.float64) + u[:, None] * e1.astype(np.float64) + v[:, None] * e2.astype(np.float64)
    return unit(p - o.astype(np.float64))


@pytest.mark.parametrize("scale,size", [(1.0, 1.0), (30.0, 0.1), (30.0, 30.0), (1000.0, 0.01)])
def test_triangle_primary_filter_never_rejects_a_reference_candidate(scale, size):
    """k_prepare_primary's DevTriF + tri2_primary_filter_pk vs ray_triangle.h's numerators with
    edges, tvec and qvec hoisted (all rays leave one origin per triangle here)."""
    rng = np.random.default_rng(int(scale * 10 + size * 100))
    n = 1_000_000
    o = (rng.uniform(-1, 1, (n, 3)) * scale).astype(f32)
    v0 = (rng.uniform(-1, 1, (n, 3)) * scale).astype(f32)
    e1 = (rng.normal(size=(n, 3)) * size).astype(f32)
    e2 = (rng.normal(size=(n, 3)) * size).astype(f32)
    d = rays_near_edges(rng, o, v0, e1, e2, n)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    tv = [f32(o[:, i] - v0[:, i]) for i in range(3)]
    qv = ref_cross(*tv, e1[:, 0], e1[:, 1], e1[:, 2])                     # ray_triangle.h:37
    # reference numerators (ray_triangle.h:18,21,32,40)
    pv = ref_cross(dx, dy, dz, e2[:, 0], e2[:, 1], e2[:, 2])
    det = ref_dot(e1[:, 0], e1[:, 1], e1[:, 2], *pv)
    un = ref_dot(*tv, *pv)
    vn = ref_dot(*qv, dx, dy, dz)
    ref_ok = uv_accept(det, un, vn)
    # filter record (k_prepare_primary) and evaluation (tri2_primary_filter_pk)
    n1 = ref_cross(e2[:, 0], e2[:, 1], e2[:, 2], e1[:, 0], e1[:, 1], e1[:, 2])
    n2 = ref_cross(e2[:, 0], e2[:, 1], e2[:, 2], *tv)
    a1, a2, at, aq = l1(*e1.T), l1(*e2.T), l1(*tv), l1(*qv)
    p12 = f32(a1 * a2)
    M = f32(f32(f32(p12 * f32(f32(p12 + f32(at * a2)) + aq)) * f32(2.0 ** -17)) + f32(2.0 ** -120))
    detf, unf, vnf = fdot(*n1, dx, dy, dz), fdot(*n2, dx, dy, dz), fdot(*qv, dx, dy, dz)
    s = f32(unf + vnf)
    A, B = fma(unf, detf, M), fma(vnf, detf, M)
    C = fma(detf, f32(detf - s), M)
    filt_ok = (A >= 0) & (B >= 0) & (C >= 0)
    assert ref_ok.sum() > n // 8 and (~ref_ok).sum() > n // 8
    missed = ref_ok & ~filt_ok
    assert not missed.any(), f"{int(missed.sum())} reference candidates rejected by the filter"
    dr = unit(rng.normal(size=(n, 3)))  # and it filters: random directions mostly fail
    detf, unf, vnf = fdot(*n1, *dr.T), fdot(*n2, *dr.T), fdot(*qv, *dr.T)
    ok = (fma(unf, detf, M) >= 0) & (fma(vnf, detf, M) >= 0) & \
         (fma(detf, f32(detf - f32(unf + vnf)), M) >= 0)
    assert ok.mean() < 0.5


@pytest.mark.parametrize("scale,size,offset", [(1.0, 1.0, 0.0), (30.0, 0.1, 0.0), (30.0, 5.0, 800.0),
                                               (1000.0, 0.01, 0.0)])
def test_triangle_shadow_filter_never_rejects_a_reference_candidate(scale, size, offset):
    """commit()'s DevTriPairF + make_ray_tri_filter + tripair2_any_filter_pk vs ray_triangle.h's
    numerators for arbitrary origins (test_tri_any), rays inside rho_max."""
    rng = np.random.default_rng(int(scale + size * 10 + offset))
    n = 1_000_000
    o = (rng.uniform(-1, 1, (n, 3)) * scale + offset).astype(f32)
    v0 = (rng.uniform(-1, 1, (n, 3)) * scale + offset).astype(f32)
    e1 = (rng.normal(size=(n, 3)) * size).astype(f32)
    e2 = (rng.normal(size=(n, 3)) * size).astype(f32)
    L = rays_near_edges(rng, o, v0, e1, e2, n)
    Lx, Ly, Lz = L[:, 0], L[:, 1], L[:, 2]
    # reference numerators (test_tri_any order)
    pv = ref_cross(Lx, Ly, Lz, e2[:, 0], e2[:, 1], e2[:, 2])
    det = ref_dot(e1[:, 0], e1[:, 1], e1[:, 2], *pv)
    tv = [f32(o[:, i] - v0[:, i]) for i in range(3)]
    un = ref_dot(*tv, *pv)
    qv = ref_cross(*tv, e1[:, 0], e1[:, 1], e1[:, 2])
    vn = ref_dot(Lx, Ly, Lz, *qv)
    ref_ok = uv_accept(det, un, vn)
    # host side (commit()): g, rho_max, record in double -> fp32, M rounded up
    pts = np.concatenate([v0, v0 + e1, v0 + e2]).astype(np.float64)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    g = (0.5 * (lo + hi)).astype(f32)
    rho = 2.0 * np.maximum(hi - g, g - lo).sum() + 1e-30
    vd = (v0.astype(np.float64) - g).astype(f32).astype(np.float64)
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    n1 = np.cross(e2d, e1d).astype(f32)
    k1 = np.cross(e1d, vd).astype(f32)
    k2 = np.cross(e2d, vd).astype(f32)
    a1, a2, av = np.abs(e1d).sum(1), np.abs(e2d).sum(1), np.abs(vd).sum(1)
    p12 = a1 * a2
    Md = 2.0 ** -17 * p12 * (p12 + (a1 + a2) * (av + rho)) + 2.0 ** -120
    M = Md.astype(f32)
    low = M.astype(np.float64) < Md
    M[low] = np.nextafter(M[low], f32(np.inf))
    # device side: a = o - g, m = a x L (un-fused), then the FMA chains
    a = [f32(o[:, i] - g[i]) for i in range(3)]
    assert float((np.abs(a[0]) + np.abs(a[1]) + np.abs(a[2])).max()) <= rho  # not `far`
    m = ref_cross(*a, Lx, Ly, Lz)
    detf = fdot(*n1.T, Lx, Ly, Lz)
    x = fdot(*k2.T, Lx, Ly, Lz)
    y = fdot(*k1.T, Lx, Ly, Lz)
    unf = fma(e2[:, 2], m[2], fma(e2[:, 1], m[1], fma(e2[:, 0], m[0], -x)))
    vnf = fma(-e1[:, 2], m[2], fma(-e1[:, 1], m[1], fma(-e1[:, 0], m[0], y)))
    s = f32(unf + vnf)
    A, B = fma(unf, detf, M), fma(vnf, detf, M)
    C = fma(detf, f32(detf - s), M)
    filt_ok = (A >= 0) & (B >= 0) & (C >= 0)
    assert ref_ok.sum() > n // 8 and (~ref_ok).sum() > n // 8
    missed = ref_ok & ~filt_ok
    assert not missed.any(), f"{int(missed.sum())} reference candidates rejected by the filter"


# ------------------------------------------------------------------ triangle pre-filters
def _tri_geometry(e1, e2):
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    s3 = (e1d + e2d) / 3.0
    rad = np.sqrt(np.maximum((s3 ** 2).sum(1), np.maximum(((e1d - s3) ** 2).sum(1),
                                                          ((e2d - s3) ** 2).sum(1))))
    emax = np.sqrt(np.maximum((e1d ** 2).sum(1), (e2d ** 2).sum(1)))
    return s3, rad, emax


@pytest.mark.parametrize("scale,size", [(1.0, 1.0), (30.0, 0.1), (30.0, 30.0), (1000.0, 0.01)])
def test_triangle_primary_prefilter_never_rejects_a_reference_candidate(scale, size):
    """k_prepare_primary's DevTriPF + tri4_primary_prefilter_pk: bounding sphere in scaled form
    OR nearly parallel.  Mirrors the fp32 statements of the kernel."""
    rng = np.random.default_rng(int(scale * 7 + size * 13))
    n = 1_000_000
    o = (rng.uniform(-1, 1, (n, 3)) * scale).astype(f32)
    v0 = (rng.uniform(-1, 1, (n, 3)) * scale).astype(f32)
    e1 = (rng.normal(size=(n, 3)) * size).astype(f32)
    e2 = (rng.normal(size=(n, 3)) * size).astype(f32)
    # a third of the triangles are thin, a third of the rays nearly in-plane
    thin = rng.uniform(size=n) < 0.33
    e2 = np.where(thin[:, None], (e1 * rng.uniform(0.5, 2, (n, 1)) +
                                  e2 * 10.0 ** rng.uniform(-4, -1, (n, 1))).astype(f32), e2)
    d = rays_near_edges(rng, o, v0, e1, e2, n)
    graze = rng.uniform(size=n) < 0.33
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    dd = d.astype(np.float64)
    dd = dd - nrm * (dd * nrm).sum(1, keepdims=True) * (1 - 10.0 ** rng.uniform(-6, -1, (n, 1)))
    d = np.where(graze[:, None], unit(dd), d)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    tv = [f32(o[:, i] - v0[:, i]) for i in range(3)]
    qv = ref_cross(*tv, e1[:, 0], e1[:, 1], e1[:, 2])
    pv = ref_cross(dx, dy, dz, e2[:, 0], e2[:, 1], e2[:, 2])
    det = ref_dot(e1[:, 0], e1[:, 1], e1[:, 2], *pv)
    un = ref_dot(*tv, *pv)
    vn = ref_dot(*qv, dx, dy, dz)
    ref_ok = uv_accept(det, un, vn)
    # k_prepare_primary, fp32 statement by statement
    n1 = ref_cross(e2[:, 0], e2[:, 1], e2[:, 2], e1[:, 0], e1[:, 1], e1[:, 2])
    a1, a2, at = l1(*e1.T), l1(*e2.T), l1(*tv)
    p12 = f32(a1 * a2)
    third = f32(1.0) / f32(3.0)
    s3 = [f32(f32(e1[:, i] + e2[:, i]) * third) for i in range(3)]
    ocg = [f32(tv[i] - s3[i]) for i in range(3)]

    def d3(a, b):
        return ref_dot(a[0], a[1], a[2], b[0], b[1], b[2])
    e1s = [f32(e1[:, i] - s3[i]) for i in range(3)]
    e2s = [f32(e2[:, i] - s3[i]) for i in range(3)]
    rho = f32(np.sqrt(np.maximum(np.maximum(d3(s3, s3), d3(e1s, e1s)), d3(e2s, e2s)), dtype=f32) * f32(1.00001))
    emax = f32(np.sqrt(np.maximum(d3(e1.T, e1.T), d3(e2.T, e2.T)), dtype=f32) * f32(1.00001))
    ok_shape = rho > f32(2.0 ** -10) * emax
    with np.errstate(all="ignore"):
        tau = f32(f32(f32(f32(3.2) * f32(2.0 ** -24)) * f32(f32(f32(f32(10.04) * at) * a2 + f32(f32(5.04) * at) * a1) + f32(20.1) * p12)) * emax / rho)
        taup = f32(f32(tau + f32(f32(10.125) * f32(2.0 ** -24)) * p12) * f32(1.00001) + f32(2.0 ** -120))
        R = f32(f32(2) * rho + f32(2.0 ** -21) * f32(f32(at + a1) + a2))
        A = l1(*ocg)
        R2 = f32(f32(R * R) * f32(1.00001))
        ccg = f32(d3(ocg, ocg) - R2)
        ccm = f32(ccg - f32(f32(f32(A * A) + R2) * f32(2.0 ** -19) + f32(2.0 ** -120)))
        sq = np.sqrt(np.maximum(ccm, f32(0)), dtype=f32)
        sc = f32(f32(f32(sq * f32(1 - 2.0 ** -22)) - f32(A * f32(9 * 2.0 ** -24))) * f32(1 - 2.0 ** -22))
        ok = ok_shape & (ccm > 0) & (sc > 0)
        inv = f32(f32(1) / np.where(ok, sc, f32(1)))
        ig = f32(f32(1) / np.where(ok_shape, taup, f32(1)))
    sx = np.where(ok, f32(ocg[0] * inv), f32(0))
    sy = np.where(ok, f32(ocg[1] * inv), f32(0))
    sz = np.where(ok, f32(ocg[2] * inv), f32(0))
    w = np.where(ok, f32(0), f32(2))
    gx = np.where(ok_shape, f32(n1[0] * ig), f32(0))
    gy = np.where(ok_shape, f32(n1[1] * ig), f32(0))
    gz = np.where(ok_shape, f32(n1[2] * ig), f32(0))
    b = fma(sz, dz, fma(sy, dy, fma(sx, dx, w)))
    gg = fma(gz, dz, fma(gy, dy, f32(gx * dx)))
    passed = (np.abs(b) >= 1) | (np.abs(gg) <= 1)
    assert ref_ok.sum() > n // 16
    missed = ref_ok & ~passed
    assert not missed.any(), f"{int(missed.sum())} reference candidates rejected by the pre-filter"
    # it filters: random directions mostly fail for small triangles -- unless they are so far away
    # (|tv| / |e| ~ 1e5) that the reference's own (u, v) noise spans them: then nearly every ray is
    # "nearly parallel" by the bound, correctly
    dr = unit(rng.normal(size=(n, 3)))
    if size <= 0.1 * scale and scale / size <= 1000:
        br = fma(sz, dr[:, 2], fma(sy, dr[:, 1], fma(sx, dr[:, 0], w)))
        gr = fma(gz, dr[:, 2], fma(gy, dr[:, 1], f32(gx * dr[:, 0])))
        assert ((np.abs(br) >= 1) | (np.abs(gr) <= 1))[~thin].mean() < 0.2


@pytest.mark.parametrize("scale,size,offset", [(1.0, 1.0, 0.0), (30.0, 0.1, 0.0), (30.0, 5.0, 800.0)])
def test_triangle_shadow_prefilter_never_rejects_a_reference_candidate(scale, size, offset):
    """commit()'s DevTriPairPF + make_ray_filter + tripair2_any_prefilter_pk: bounding sphere in
    the shadow-sphere filter's form OR nearly parallel, for arbitrary origins inside rho_max."""
    rng = np.random.default_rng(int(scale * 3 + size * 17 + offset))
    n = 1_000_000
    o = (rng.uniform(-1, 1, (n, 3)) * scale + offset).astype(f32)
    v0 = (rng.uniform(-1, 1, (n, 3)) * scale + offset).astype(f32)
    e1 = (rng.normal(size=(n, 3)) * size).astype(f32)
    e2 = (rng.normal(size=(n, 3)) * size).astype(f32)
    thin = rng.uniform(size=n) < 0.33
    e2 = np.where(thin[:, None], (e1 * rng.uniform(0.5, 2, (n, 1)) +
                                  e2 * 10.0 ** rng.uniform(-4, -1, (n, 1))).astype(f32), e2)
    L = rays_near_edges(rng, o, v0, e1, e2, n)
    graze = rng.uniform(size=n) < 0.33
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    dd = L.astype(np.float64)
    dd = dd - nrm * (dd * nrm).sum(1, keepdims=True) * (1 - 10.0 ** rng.uniform(-6, -1, (n, 1)))
    L = np.where(graze[:, None], unit(dd), L)
    Lx, Ly, Lz = L[:, 0], L[:, 1], L[:, 2]
    pv = ref_cross(Lx, Ly, Lz, e2[:, 0], e2[:, 1], e2[:, 2])
    det = ref_dot(e1[:, 0], e1[:, 1], e1[:, 2], *pv)
    tv = [f32(o[:, i] - v0[:, i]) for i in range(3)]
    un = ref_dot(*tv, *pv)
    qv = ref_cross(*tv, e1[:, 0], e1[:, 1], e1[:, 2])
    vn = ref_dot(Lx, Ly, Lz, *qv)
    ref_ok = uv_accept(det, un, vn)
    # host side (commit()), double
    pts = np.concatenate([v0, v0 + e1, v0 + e2]).astype(np.float64)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    g = (0.5 * (lo + hi)).astype(f32)
    rho = 2.0 * np.maximum(hi - g, g - lo).sum() + 1e-30
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    s3, rad, emax = _tri_geometry(e1, e2)
    G = v0.astype(np.float64) + s3
    a1, a2 = np.abs(e1d).sum(1), np.abs(e2d).sum(1)
    av = np.abs((v0.astype(np.float64) - g).astype(f32).astype(np.float64)).sum(1)
    u = 2.0 ** -24
    at, p12 = rho + av, a1 * a2
    ok_shape = rad > 2.0 ** -10 * emax
    with np.errstate(all="ignore"):
        tau = 3.2 * u * (10.04 * at * a2 + 5.04 * at * a1 + 20.1 * p12) * emax / rad
    taup = (tau + 10.1 * u * p12) * 1.00001 + 2.0 ** -120
    R = 2.0 * rad + 8.0 * u * (at + a1 + a2)
    c = (G - g).astype(f32)
    c2 = (c.astype(np.float64) ** 2).sum(1)
    R2 = R * R * 1.00001
    km_d = R2 - c2 + 2.0 ** -16 * (c2 + R2) + 2.0 ** -120
    km = km_d.astype(f32)
    low = km.astype(np.float64) < km_d
    km[low] = np.nextafter(km[low], f32(np.inf))
    km = np.where(ok_shape, km, f32(np.inf))
    n1 = np.cross(e2d, e1d)
    with np.errstate(all="ignore"):
        gv = np.where(ok_shape[:, None], (n1 / taup[:, None]), 0.0).astype(f32)
    cc = np.where(ok_shape[:, None], c, f32(0))
    # device side: make_ray_filter + tripair2_any_prefilter_pk
    ax, ay, az = f32(o[:, 0] - g[0]), f32(o[:, 1] - g[1]), f32(o[:, 2] - g[2])
    assert float((np.abs(ax) + np.abs(ay) + np.abs(az)).max()) <= rho
    nn = ref_dot(ax, ay, az, ax, ay, az)
    ss = ref_dot(ax, ay, az, Lx, Ly, Lz)
    nko = f32(nn * f32(-(1.0 - 2.0 ** -16)))
    y = fma(cc[:, 2], f32(az + az), fma(cc[:, 1], f32(ay + ay), fma(cc[:, 0], f32(ax + ax), nko)))
    x = fma(cc[:, 2], Lz, fma(cc[:, 1], Ly, fma(cc[:, 0], Lx, f32(-ss))))
    with np.errstate(all="ignore"):
        q = f32(fma(x, x, y) + km)
    gg = fma(gv[:, 2], Lz, fma(gv[:, 1], Ly, f32(gv[:, 0] * Lx)))
    passed = (q >= 0) | (np.abs(gg) <= 1)
    assert ref_ok.sum() > n // 16
    missed = ref_ok & ~passed
    assert not missed.any(), f"{int(missed.sum())} reference candidates rejected by the pre-filter"


# ------------------------------------------------------------------ triangle groups
def _escape_possible(o, v0, e1, e2):
    """numpy mirror of k_prepare_tri_groups' tri_escape (fp32), one camera per row: possible,
    bounded (unit normal and beta are usable), unit normal, beta = tau / |n1|"""
    u = f32(2.0 ** -24)
    tv = [f32(o[:, i] - v0[:, i]) for i in range(3)]
    n1 = ref_cross(e2[:, 0], e2[:, 1], e2[:, 2], e1[:, 0], e1[:, 1], e1[:, 2])

    def d3(a, b):
        return ref_dot(a[0], a[1], a[2], b[0], b[1], b[2])
    nn = np.sqrt(d3(n1, n1), dtype=f32)
    a1, a2, at = l1(*e1.T), l1(*e2.T), l1(*tv)
    len1, len2 = np.sqrt(d3(e1.T, e1.T), dtype=f32), np.sqrt(d3(e2.T, e2.T), dtype=f32)
    third = f32(1.0) / f32(3.0)
    s3 = [f32(f32(e1[:, i] + e2[:, i]) * third) for i in range(3)]
    e1s = [f32(e1[:, i] - s3[i]) for i in range(3)]
    e2s = [f32(e2[:, i] - s3[i]) for i in range(3)]
    rho = np.sqrt(np.maximum(np.maximum(d3(s3, s3), d3(e1s, e1s)), d3(e2s, e2s)), dtype=f32)
    emax = np.maximum(len1, len2)
    with np.errstate(all="ignore"):
        yes = ~(rho > f32(1.125 * 2.0 ** -10) * emax) | ~(nn > 0)
        p12 = f32(a1 * a2)
        k = f32(f32(f32(f32(3.2) * u) * emax) / rho * f32(1.0001))
        tau = f32(k * f32(f32(f32(f32(10.04) * a2) + f32(f32(5.04) * a1)) * at + f32(f32(20.1) * p12)))
        yes |= ~(tau < f32(f32(0.1) * nn))
        ted = f32(f32(tau + f32(f32(f32(10.05) * u) * p12)) * f32(f32(1) + f32(4) * u))
        U = f32(ted + f32(f32(f32(f32(10.04) * u) * at) * a2))
        V = f32(ted + f32(f32(f32(f32(5.04) * u) * at) * a1))
        H = f32(f32(f32(np.maximum(f32(V / len1), f32(U / len2)) + f32(f32(at * tau) / nn)) *
                    f32(f32(f32(f32(2) * len1) * len2) / nn)) * f32(f32(1) / f32(0.99)))
        h = f32(np.abs(d3(tv, n1)) / nn)
        bounded = ~yes
        yes |= ~(h > f32(f32(H * f32(1.01)) + f32(f32(2.0 ** -20) * at)))
        inv = f32(f32(1) / nn)
        nh = np.stack([f32(n1[0] * inv), f32(n1[1] * inv), f32(n1[2] * inv)], axis=1)
        beta = f32(tau / nn)
    return yes, bounded, nh, beta


def _patch_groups(rng, n_grp, scale, size, offset=0.0):
    """n_grp groups of 8 triangles: a 2 x 2 quad patch of y = bump(x, z), flat or curved, randomly
    rotated and placed; static group records from the library (esc_tri_group_record, host only)"""
    import ctypes as C
    from esctp1raytracer_amd import _capi
    lib = _capi.load()
    gx, gz = np.meshgrid(np.arange(3), np.arange(3), indexing="ij")
    rec = np.zeros((n_grp, 12), f32)
    V0 = np.zeros((n_grp, 8, 3), f32)
    E1 = np.zeros((n_grp, 8, 3), f32)
    E2 = np.zeros((n_grp, 8, 3), f32)
    for g in range(n_grp):
        ph = rng.uniform(0, 6.3, 2)
        amp = rng.choice([0.0, 0.05, 0.5])  # flat groups (coplanar members) included
        P = np.stack([gx * size, amp * size * np.sin(gx * 0.8 + ph[0]) * np.cos(gz * 0.7 + ph[1]), gz * size], -1)
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        P = P @ Q.T + rng.uniform(-1, 1, 3) * scale + offset
        a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
        tri = np.concatenate([np.stack([a, b, c], 2).reshape(-1, 3, 3), np.stack([a, c, d], 2).reshape(-1, 3, 3)])
        tri = tri.astype(f32)
        V0[g], E1[g], E2[g] = tri[:, 0], f32(tri[:, 1] - tri[:, 0]), f32(tri[:, 2] - tri[:, 0])
        buf = np.ascontiguousarray(np.concatenate([V0[g], E1[g], E2[g]], axis=1), f32)
        assert lib.esc_tri_group_record(buf.ctypes.data_as(C.POINTER(C.c_float)), 8,
                                        rec[g].ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert (rec[:, 11] == 0).all()  # none is `always`
    return rec, V0, E1, E2


@pytest.mark.parametrize("scale,size,with_slab", [(1.0, 0.3, True), (30.0, 0.1, True), (30.0, 3.0, True),
                                                  (1000.0, 0.05, True), (30.0, 3.0, False)])
def test_triangle_primary_group_never_rejects_a_member_candidate(scale, size, with_slab):
    """rt_device.h TriGroups, primary rays: the group record (bounding sphere of the members'
    pre-filter spheres OR "nearly parallel" to the frame's cone -- built over the members whose
    plane the camera is within H_t of, statement (P) of rt_brute.h; none: no cone) against the
    reference accept of a member.  Groups of 8 triangles off a bumpy patch; cameras anywhere and
    (half of them) a hair off a member's plane, rays that hug the member's edges or run nearly in
    its plane.  Static bounds come from the library (esc_tri_group_record, host only); the
    per-frame record is mirrored here statement by statement.  with_slab=False drops (P)'s
    condition and switches every cone off: the same rays must then lose candidates -- the
    scenario has the power to see the escape missing (with edges of a few units: for small
    triangles a det of rounding noise, ~u |e1||e2|, never clears the reference's absolute
    |det| > FLT_EPSILON)."""
    rng = np.random.default_rng(int(scale * 3 + size * 100))
    n_grp, per = 6000, 60
    rec, V0, E1, E2 = _patch_groups(rng, n_grp, scale, size)
    n = n_grp * per
    gi = np.repeat(np.arange(n_grp), per)
    mi = rng.integers(0, 8, n)
    v0, e1, e2 = V0[gi, mi], E1[gi, mi], E2[gi, mi]
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    # cameras: random; half of them moved to within a hair of the member's plane
    o = rng.uniform(-1, 1, (n, 3)) * scale * 1.5
    inplane = rng.uniform(size=n) < 0.5
    hgt = ((o - v0) * nrm).sum(1, keepdims=True)
    dist = np.linalg.norm(o - v0, axis=1, keepdims=True)
    o = np.where(inplane[:, None], o - nrm * hgt + nrm * dist * rng.choice([-1, 1], (n, 1)) *
                 10.0 ** rng.uniform(-9, -3, (n, 1)), o).astype(f32)
    d = rays_near_edges(rng, o, v0, e1, e2, n)
    graze = rng.uniform(size=n) < 0.5
    dd = d.astype(np.float64)
    dd = dd - nrm * (dd * nrm).sum(1, keepdims=True) * (1 - 10.0 ** rng.uniform(-9, -2, (n, 1)))
    d = np.where(graze[:, None], unit(dd), d)
    # a third of the rays: camera AND ray in the member's plane, close by, the line passing the
    # whole group at 1.2 .. 3 of its radii -- only the escape can let these through, and the
    # reference does accept some of them (everything it computes there is rounding noise)
    G = rec[gi]
    Cg, rg = G[:, 0:3].astype(np.float64), G[:, 3:4].astype(np.float64)
    miss = rng.uniform(size=n) < 0.34
    t1 = np.cross(nrm, rng.normal(size=(n, 3)))
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    Cp = Cg - nrm * ((Cg - v0) * nrm).sum(1, keepdims=True)       # C projected into the plane
    om = Cp + t1 * rg * rng.uniform(3, 10, (n, 1))                 # camera in the plane
    aim = Cp + t2 * rg * rng.uniform(1.2, 3, (n, 1)) * rng.choice([-1, 1], (n, 1))
    o = np.where(miss[:, None], om.astype(f32), o)
    d = np.where(miss[:, None], unit(aim - o.astype(np.float64)), d)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    # reference accept of the member
    tv = [f32(o[:, i] - v0[:, i]) for i in range(3)]
    qv = ref_cross(*tv, e1[:, 0], e1[:, 1], e1[:, 2])
    pv = ref_cross(dx, dy, dz, e2[:, 0], e2[:, 1], e2[:, 2])
    det = ref_dot(e1[:, 0], e1[:, 1], e1[:, 2], *pv)
    ref_ok = uv_accept(det, ref_dot(*tv, *pv), ref_dot(*qv, dx, dy, dz))
    # k_prepare_tri_groups, fp32 statement by statement
    oc = [f32(o[:, i] - G[:, i]) for i in range(3)]
    A = l1(*oc)
    at = f32(A + G[:, 8])
    R = f32(f32(G[:, 3] + f32(f32(2.0 ** -21) * at)) + f32(2.0 ** -60))
    R2 = f32(f32(R * R) * f32(1.00001))
    sx, sy, sz, w = _scaled_record(oc[0], oc[1], oc[2], f32(ref_dot(*oc, *oc) - R2), R2)
    # the frame's cone: over the members whose plane the camera is within H_t of, only
    acc = np.zeros((n, 3), f32)
    ref = np.zeros((n, 3), f32)
    bmax = np.zeros(n, f32)
    cnt = np.zeros(n, np.int32)
    unbounded = np.zeros(n, bool)
    flagged, normals = [], []
    for m in range(8):
        pos, bnd, nh, beta = _escape_possible(o, V0[gi, m], E1[gi, m], E2[gi, m])
        if not with_slab:
            pos = np.zeros(n, bool)
        unbounded |= pos & ~bnd
        use = pos & bnd
        first = use & (cnt == 0)
        ref = np.where(first[:, None], nh, ref)
        sgn = np.where(ref_dot(nh[:, 0], nh[:, 1], nh[:, 2], ref[:, 0], ref[:, 1], ref[:, 2]) < 0, f32(-1), f32(1))
        acc = np.where(use[:, None], f32(acc + f32(nh * sgn[:, None])), acc)
        bmax = np.where(use, np.maximum(bmax, beta), bmax)
        cnt += use
        flagged.append(use)
        normals.append(nh)
    possible = unbounded | (cnt > 0)
    with np.errstate(all="ignore"):
        an = np.sqrt(ref_dot(acc[:, 0], acc[:, 1], acc[:, 2], acc[:, 0], acc[:, 1], acc[:, 2]), dtype=f32)
        unbounded |= (cnt > 0) & ~(an > f32(0.5) * cnt.astype(f32))
        ax = f32(acc * f32(f32(1) / an)[:, None])
        smax = np.zeros(n, f32)
        for m in range(8):
            c = ref_cross(ax[:, 0], ax[:, 1], ax[:, 2], normals[m][:, 0], normals[m][:, 1], normals[m][:, 2])
            sm = np.sqrt(ref_dot(c[0], c[1], c[2], c[0], c[1], c[2]), dtype=f32)
            smax = np.where(flagged[m], np.maximum(smax, sm), smax)
        kp = f32(f32(f32(f32(f32(smax + f32(1e-5)) * f32(1.0001)) + bmax) + f32(2.0 ** -20)) * f32(1.0001))
        cone = ~unbounded & (cnt > 0) & (kp < 1)
        ik = f32(f32(1) / kp)
    never = ~unbounded & (cnt == 0)
    gxv = np.where(cone, f32(ax[:, 0] * ik), np.where(never, f32(2.0 ** 60), f32(0)))
    gyv = np.where(cone, f32(ax[:, 1] * ik), f32(0))
    gzv = np.where(cone, f32(ax[:, 2] * ik), f32(0))
    b = fma(sz, dz, fma(sy, dy, fma(sx, dx, w)))
    gg = fma(gzv, dz, fma(gyv, dy, f32(gxv * dx)))
    opened = (np.abs(b) >= 1) | (np.abs(gg) <= 1)
    assert ref_ok.sum() > n // 20 and (~ref_ok).sum() > n // 20
    missed = ref_ok & ~opened
    if with_slab:
        assert not missed.any(), f"{int(missed.sum())} member accepts behind a closed group"
        # and the cone is off for most cameras that are NOT near a member's plane (unless the
        # triangles are so small for their distance that every accept is rounding noise)
        if scale / size < 1000:
            assert possible[~inplane & ~miss].mean() < 0.2
    else:
        assert missed.any()


@pytest.mark.parametrize("scale,size,offset,cone", [(1.0, 0.3, 0.0, True), (30.0, 0.1, 0.0, True),
                                                    (30.0, 3.0, 0.0, True), (30.0, 3.0, 400.0, True),
                                                    (30.0, 3.0, 0.0, False)])
def test_triangle_shadow_group_never_rejects_a_member_candidate(scale, size, offset, cone):
    """rt_device.h TriGroups, shadow rays: commit()'s group record in DevTriPairPF form (bounding
    sphere with R = rgeo + 8u at in the q' form, cone axis over kappa' = (smax + b0 + b1 at +
    2^-20) 1.0001, at = rho_max + |C - g|_1 + rext) against the reference's any-hit accept of a
    member, for origins inside the scene box -- on member planes and off them -- and rays that
    hug member edges, run nearly in a member's plane, or lie IN it and pass the whole group by.
    cone=False drops the cone part: the same rays must then lose candidates."""
    rng = np.random.default_rng(int(scale * 5 + size * 70 + offset))
    n_grp, per = 5000, 60
    rec, V0, E1, E2 = _patch_groups(rng, n_grp, scale, size, offset)
    n = n_grp * per
    gi = np.repeat(np.arange(n_grp), per)
    mi = rng.integers(0, 8, n)
    v0, e1, e2 = V0[gi, mi], E1[gi, mi], E2[gi, mi]
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    G = rec[gi]
    Cg, rg = G[:, 0:3].astype(np.float64), G[:, 3:4].astype(np.float64)
    o = rng.uniform(-1, 1, (n, 3)) * scale + offset
    inplane = rng.uniform(size=n) < 0.5
    hgt = ((o - v0) * nrm).sum(1, keepdims=True)
    dist = np.linalg.norm(o - v0, axis=1, keepdims=True)
    o = np.where(inplane[:, None], o - nrm * hgt + nrm * dist * rng.choice([-1, 1], (n, 1)) *
                 10.0 ** rng.uniform(-9, -3, (n, 1)), o).astype(f32)
    L = rays_near_edges(rng, o, v0, e1, e2, n)
    graze = rng.uniform(size=n) < 0.5
    dd = L.astype(np.float64)
    dd = dd - nrm * (dd * nrm).sum(1, keepdims=True) * (1 - 10.0 ** rng.uniform(-9, -2, (n, 1)))
    L = np.where(graze[:, None], unit(dd), L)
    miss = rng.uniform(size=n) < 0.34  # origin and ray in the member's plane, passing the group by
    t1 = np.cross(nrm, rng.normal(size=(n, 3)))
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    Cp = Cg - nrm * ((Cg - v0) * nrm).sum(1, keepdims=True)
    om = Cp + t1 * rg * rng.uniform(3, 10, (n, 1))
    aim = Cp + t2 * rg * rng.uniform(1.2, 3, (n, 1)) * rng.choice([-1, 1], (n, 1))
    o = np.where(miss[:, None], om.astype(f32), o)
    L = np.where(miss[:, None], unit(aim - o.astype(np.float64)), L)
    Lx, Ly, Lz = L[:, 0], L[:, 1], L[:, 2]
    # reference any-hit numerators (test_tri_any)
    pv = ref_cross(Lx, Ly, Lz, e2[:, 0], e2[:, 1], e2[:, 2])
    det = ref_dot(e1[:, 0], e1[:, 1], e1[:, 2], *pv)
    tv = [f32(o[:, i] - v0[:, i]) for i in range(3)]
    qv = ref_cross(*tv, e1[:, 0], e1[:, 1], e1[:, 2])
    ref_ok = uv_accept(det, ref_dot(*tv, *pv), ref_dot(Lx, Ly, Lz, *qv))
    # host side (commit()): g, rho_max over everything; the group record in double
    pts = np.concatenate([V0.reshape(-1, 3), (V0 + E1).reshape(-1, 3), (V0 + E2).reshape(-1, 3),
                          o]).astype(np.float64)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    g = (0.5 * (lo + hi)).astype(f32)
    rho = 2.0 * np.maximum(hi - g, g - lo).sum() + 1e-30
    c1 = np.abs(Cg - g.astype(np.float64)).sum(1)
    at = rho + c1 + G[:, 8].astype(np.float64)
    kappa = (G[:, 7].astype(np.float64) + G[:, 9] + G[:, 10].astype(np.float64) * at + 2.0 ** -20) * 1.0001
    usable = kappa < 1.0
    R = rg[:, 0] + 2.0 ** -21 * at + 2.0 ** -60
    c = (Cg - g.astype(np.float64)).astype(f32)
    c2 = (c.astype(np.float64) ** 2).sum(1)
    R2 = R * R * 1.00001
    km_d = R2 - c2 + 2.0 ** -16 * (c2 + R2) + 2.0 ** -120
    km = km_d.astype(f32)
    low = km.astype(np.float64) < km_d
    km[low] = np.nextafter(km[low], f32(np.inf))
    km = np.where(usable, km, f32(np.inf))
    gv = np.where(usable[:, None], G[:, 4:7].astype(np.float64) / kappa[:, None], 0.0).astype(f32)
    if not cone:
        gv = np.full_like(gv, f32(2.0 ** 60))
    cc = np.where(usable[:, None], c, f32(0))
    # device side: make_ray_filter + tripair2_any_prefilter_pk
    ax, ay, az = f32(o[:, 0] - g[0]), f32(o[:, 1] - g[1]), f32(o[:, 2] - g[2])
    assert float((np.abs(ax) + np.abs(ay) + np.abs(az)).max()) <= rho  # none is `far`
    nn = ref_dot(ax, ay, az, ax, ay, az)
    ss = ref_dot(ax, ay, az, Lx, Ly, Lz)
    nko = f32(nn * f32(-(1.0 - 2.0 ** -16)))
    y = fma(cc[:, 2], f32(az + az), fma(cc[:, 1], f32(ay + ay), fma(cc[:, 0], f32(ax + ax), nko)))
    x = fma(cc[:, 2], Lz, fma(cc[:, 1], Ly, fma(cc[:, 0], Lx, f32(-ss))))
    with np.errstate(all="ignore"):
        q = f32(fma(x, x, y) + km)
        gg = fma(gv[:, 2], Lz, fma(gv[:, 1], Ly, f32(gv[:, 0] * Lx)))
    opened = (q >= 0) | (np.abs(gg) <= 1)
    assert ref_ok.sum() > n // 20 and (~ref_ok).sum() > n // 20
    missed = ref_ok & ~opened
    if cone:
        assert not missed.any(), f"{int(missed.sum())} member accepts behind a closed group"
    else:
        assert missed.any()
