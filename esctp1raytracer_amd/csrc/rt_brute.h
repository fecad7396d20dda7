// rt_brute.h -- the brute-force primitive loops (main.cpp:176-192 closest hit, :314-329 any hit)
// over wave-uniform records: software-pipelined fetches, hand-scheduled packed-fp32 sphere bodies,
// and the re-packing of undecided shadow rays inside a workgroup.  Included by rt_kernels.hip only.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"
#include "rt_math.h"

namespace esc {
// ---------------------------------------------------------------------------------------
// primitive loops.  `rec(k)` yields record k wave-uniformly (SGPRs or LDS broadcast).
//
// Every lane carries PX pixels (same row, 16 columns apart).  A wave therefore amortises each
// primitive fetch, each wave-uniform branch and each s_waitcnt over PX x 64 rays instead of
// 64: at PX = 1 the rocprofv3 counters showed the VALU pipe 66 % busy with a quarter of all
// wave-cycles parked on scalar-load waits (profiles/r01_c4_1gpu); the arithmetic per ray is
// unchanged.
// ---------------------------------------------------------------------------------------

struct Hit {
  float t;     // main.cpp:715 FLT_MAX, then closest t
  float v;     // quirk S1: only v survives (main.cpp:307,310)
  int32_t idx; // -1 none; [0,n_tri) triangle; n_tri + k sphere k
};

// Every loop below is software-pipelined by hand: the records of the NEXT batch are fetched
// (s_load_dwordx8/x16, or ds_read_b128) before the current batch is tested, in two
// alternating register sets, so a wave never waits on the fetch it just issued.

template <typename Rec, int B, typename Fetch>
DEVINL void fetch_batch(Fetch rec, int k, Rec (&r)[B]) {
#pragma unroll
  for (int i = 0; i < B; ++i) r[i] = rec(k + i);
  // hipcc's scheduler otherwise sinks the fetch to just above its first use (measured in the
  // ISA: the s_load landed 4 instructions before the s_waitcnt); pin it where it is written.
  __builtin_amdgcn_sched_barrier(0);
}

// ---- closest hit, primary rays, triangles ------------------------------------------------
template <typename V, int NV>
DEVINL void test_tri2_primary(const DevTriP (&T)[2], int idx, const V3<V> (&d)[NV],
                              Hit (&h)[NV * lanes_of<V>::n]) {
  constexpr int LN = lanes_of<V>::n;
  V det[NV][2], un[NV][2], vn[NV][2];
  bool any = false;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const V3<V> pv = cross_vu(d[j], ld3(T[i].e2)); // ray_triangle.h:18
      det[j][i] = dotu(ld3(T[i].e1), pv);            // :21
      un[j][i] = dotu(ld3(T[i].tv), pv);             // :32 numerator
      vn[j][i] = dotu(ld3(T[i].qv), d[j]);           // :40 numerator (dot is commutative per term)
#pragma unroll
      for (int c = 0; c < LN; ++c)
        any |= tri_candidate(comp(det[j][i], c), comp(un[j][i], c), comp(vn[j][i], c));
    }
  if (ANY_LANE_RARE(any)) { // wave-uniform skip of the f64 tail
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int c = 0; c < LN; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          Hit &hh = h[j * LN + c];
          const float de = comp(det[j][i], c), u = comp(un[j][i], c), v = comp(vn[j][i], c);
          float t2, v2;
          if (tri_candidate(de, u, v) && tri_exact(de, u, v, T[i].tnum, hh.t, t2, v2)) {
            hh.t = t2;
            hh.v = v2;
            hh.idx = idx + i;
          }
        }
  }
}

template <typename V, int NV, typename Fetch>
DEVINL void closest_tri_primary(Fetch rec, int n, int base, const V3<V> (&d)[NV],
                                Hit (&h)[NV * lanes_of<V>::n]) {
  const int n4 = n & ~3;
  if (n4) {
    DevTriP A[2], B[2];
    fetch_batch(rec, 0, A);
    for (int k = 0; k < n4; k += 4) {
      fetch_batch(rec, rec.landed(A[1].tnum, k + 2), B);
      test_tri2_primary<V, NV>(A, base + k, d, h);
      fetch_batch(rec, rec.landed(B[1].tnum, min(k + 4, n - 2)), A); // clamped: last one unused
      test_tri2_primary<V, NV>(B, base + k + 2, d, h);
    }
  }
  for (int k = n4; k + 1 <= n; k += 1) {
    // remainder: reuse the pair body with the last record duplicated as a dead second slot
    DevTriP P[2] = {rec(k), rec(k)};
    P[1].e1[0] = P[1].e1[1] = P[1].e1[2] = 0.f; // det = 0 -> |det| < eps -> rejected
    P[1].tv[0] = P[1].tv[1] = P[1].tv[2] = 0.f;
    P[1].qv[0] = P[1].qv[1] = P[1].qv[2] = 0.f;
    test_tri2_primary<V, NV>(P, base + k, d, h);
  }
}

// ---- hand-scheduled packed-fp32 bodies (2 pixels per lane, SGPR operands) ----------------------
// hipcc's own v2f code for these tests spends a v_mov per hi-half broadcast and serialises the
// dependent v_pk chains (s_nop hazards); measured, it is no faster than scalar code.  Written by
// hand: every sphere constant is read straight from its SGPR pair through op_sel (lo or hi half
// to both lanes), the independent chains of the batch are interleaved so no v_pk result is
// consumed by the next instruction, and nothing but v_pk_mul_f32 / v_pk_add_f32 (with neg
// modifiers, which are exact) is used -- each half rounds exactly like the scalar v_mul / v_add /
// v_sub of the generic path, in the same order.
// The "any candidate?" filter works on the raw bits: a value is non-negative iff its bit
// pattern is >= 0 as a signed int (a disc of -0 cannot occur: b*b is >= +0 and x - x = +0).
struct SphP2 { // DevSphP seen as two aligned pairs: (ocx, ocy), (ocz, cc)
  v2f xy, zc;
};

DEVINL int max3i(int a, int b, int c) { return max(max(a, b), c); }
DEVINL bool any_nonneg(v2f a, v2f b, v2f c, v2f d) {
  int m = max3i(__float_as_int(a.x), __float_as_int(a.y), __float_as_int(b.x));
  m = max3i(m, __float_as_int(b.y), __float_as_int(c.x));
  m = max3i(m, __float_as_int(c.y), __float_as_int(d.x));
  m = max(m, __float_as_int(d.y));
  return m >= 0;
}

// primary rays, 4 spheres x 2 pixels: b = (ocx*dx + ocy*dy) + ocz*dz ; q = b*b - cc
DEVINL void sph4_primary_pk(const SphP2 (&s)[4], v2f dx, v2f dy, v2f dz, v2f (&b)[4], v2f (&q)[4]) {
  asm("v_pk_mul_f32 %0, %[s0a], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %1, %[s1a], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %2, %[s2a], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %3, %[s3a], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %4, %[s0a], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %5, %[s1a], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %6, %[s2a], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %7, %[s3a], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
      "v_pk_add_f32 %0, %0, %4\n\t"
      "v_pk_add_f32 %1, %1, %5\n\t"
      "v_pk_add_f32 %2, %2, %6\n\t"
      "v_pk_add_f32 %3, %3, %7\n\t"
      "v_pk_mul_f32 %4, %[s0b], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %5, %[s1b], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %6, %[s2b], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %7, %[s3b], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, %4\n\t"
      "v_pk_add_f32 %1, %1, %5\n\t"
      "v_pk_add_f32 %2, %2, %6\n\t"
      "v_pk_add_f32 %3, %3, %7\n\t"
      "v_pk_mul_f32 %4, %0, %0\n\t"
      "v_pk_mul_f32 %5, %1, %1\n\t"
      "v_pk_mul_f32 %6, %2, %2\n\t"
      "v_pk_mul_f32 %7, %3, %3\n\t"
      "v_pk_add_f32 %4, %4, %[s0b] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %5, %[s1b] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %6, %[s2b] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %7, %7, %[s3b] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 0"
      : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]),
        "=&v"(q[3])
      : [x] "v"(dx), [y] "v"(dy), [z] "v"(dz), [s0a] "s"(s[0].xy), [s0b] "s"(s[0].zc),
        [s1a] "s"(s[1].xy), [s1b] "s"(s[1].zc), [s2a] "s"(s[2].xy), [s2b] "s"(s[2].zc),
        [s3a] "s"(s[3].xy), [s3b] "s"(s[3].zc));
}

// ---- 1 ray per lane, TWO SPHERES per packed op (shadow rays) ---------------------------------
// The lane keeps one ray and the two halves of every v_pk op hold spheres 2j and 2j+1, whose
// constants arrive pair-interleaved (DevSphPair) and feed the ops as plain SGPR pairs; the ray's
// components are broadcast to both halves through op_sel.  This is the REFERENCE arithmetic of
// the shadow test (16 operations per pair, no fusion): what the filter's candidates run, and what
// every pair runs under ESC_RENDER_EXACT_ONLY.
struct PairG { // DevSphPair
  v2f x, y, z, r;
};

// shadow: oc = o - c ; b = (ocx*Lx + ocy*Ly) + ocz*Lz ; cc = ((ocx^2 + ocy^2) + ocz^2) - r2 ;
// q = b*b - cc, for records R0 (spheres 0,1) and R1 (spheres 2,3)
DEVINL void pair2_any_pk(const PairG (&R)[2], v2f oxy, v2f oz_, v2f Lxy, v2f Lz_, v2f (&b)[2],
                         v2f (&q)[2]) {
  // Scheduling rule measured in tools/ubench/valu_rate.hip (modes 8/10): a v_pk result must not
  // be consumed within the next 3 instructions of the same wave (other waves do not fill the
  // gap): 227 -> 148 cycles per block.  Four chains are kept in flight: dot(oc,L) and dot(oc,oc)
  // of record 0 and of record 1.
  v2f ax, ay, az, bx, by, bz, t0, t1, u0, u1;
  asm("v_pk_add_f32 %[ax], %[oxy], %[r0x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[bx], %[oxy], %[r1x] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ay], %[oxy], %[r0y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[by], %[oxy], %[r1y] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[az], %[oz], %[r0z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[bz], %[oz], %[r1z] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[b0], %[ax], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[b1], %[bx], %[Lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[q0], %[ax], %[ax]\n\t"
      "v_pk_mul_f32 %[q1], %[bx], %[bx]\n\t"
      "v_pk_mul_f32 %[t0], %[ay], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[t1], %[by], %[Lxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[u0], %[ay], %[ay]\n\t"
      "v_pk_mul_f32 %[u1], %[by], %[by]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[u0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[u1]\n\t"
      "v_pk_mul_f32 %[t0], %[az], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t1], %[bz], %[Lz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[u0], %[az], %[az]\n\t"
      "v_pk_mul_f32 %[u1], %[bz], %[bz]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[u0]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[u1]\n\t"
      "v_pk_mul_f32 %[t0], %[b0], %[b0]\n\t"
      "v_pk_mul_f32 %[t1], %[b1], %[b1]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[r0r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[r1r] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 1\n\t"
      "v_pk_add_f32 %[q0], %[t0], %[q0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[t1], %[q1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 0"
      : [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [q0] "=&v"(q[0]), [q1] "=&v"(q[1]), [ax] "=&v"(ax),
        [ay] "=&v"(ay), [az] "=&v"(az), [bx] "=&v"(bx), [by] "=&v"(by), [bz] "=&v"(bz),
        [t0] "=&v"(t0), [t1] "=&v"(t1), [u0] "=&v"(u0), [u1] "=&v"(u1)
      : [oxy] "v"(oxy), [oz] "v"(oz_), [Lxy] "v"(Lxy), [Lz] "v"(Lz_), [r0x] "s"(R[0].x),
        [r0y] "s"(R[0].y), [r0z] "s"(R[0].z), [r0r] "s"(R[0].r), [r1x] "s"(R[1].x),
        [r1y] "s"(R[1].y), [r1z] "s"(R[1].z), [r1r] "s"(R[1].r));
}

// ---- closest hit, primary rays, spheres ---------------------------------------------------
template <typename V, int NV, int NB>
DEVINL void test_sph_primary(const DevSphP (&s)[NB], int idx, const V3<V> (&d)[NV],
                             Hit (&h)[NV * lanes_of<V>::n]) {
  constexpr int LN = lanes_of<V>::n;
  V b[NV][NB], q[NV][NB];
  float m = -1.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      b[j][i] = (s[i].ocx * d[j].x + s[i].ocy * d[j].y) + s[i].ocz * d[j].z;
      q[j][i] = b[j][i] * b[j][i] - s[i].cc;
#pragma unroll
      for (int c = 0; c < LN; ++c) m = fmaxf(m, comp(q[j][i], c));
    }
  if (ANY_LANE_RARE(!(m < 0.f))) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int c = 0; c < LN; ++c)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          Hit &hh = h[j * LN + c];
          float t2;
          if (sph_exact(comp(b[j][i], c), comp(q[j][i], c), hh.t, t2)) {
            hh.t = t2;
            hh.idx = idx + i;
          }
        }
  }
}

template <typename V, int NV, typename Fetch>
DEVINL void closest_sph_primary(Fetch rec, int n, int base, const V3<V> (&d)[NV],
                                Hit (&h)[NV * lanes_of<V>::n]) {
  const int n8 = n & ~7;
  if (n8) {
    DevSphP A[4], B[4];
    fetch_batch(rec, 0, A);
    for (int k = 0; k < n8; k += 8) {
      fetch_batch(rec, rec.landed(A[3].cc, k + 4), B);
      test_sph_primary<V, NV, 4>(A, base + k, d, h);
      fetch_batch(rec, rec.landed(B[3].cc, min(k + 8, n - 4)), A);
      test_sph_primary<V, NV, 4>(B, base + k + 4, d, h);
    }
  }
  for (int k = n8; k < n; ++k) {
    const DevSphP s0[1] = {rec(k)};
    test_sph_primary<V, NV, 1>(s0, base + k, d, h);
  }
}

// SMEM + 2 pixels per lane: the hand-scheduled packed body above
template <typename Fetch>
DEVINL void closest_sph_primary_pk(Fetch rec, int n, int base, const V3<v2f> &d, Hit (&h)[2]) {
  auto test4 = [&](const SphP2(&S)[4], int idx) {
    v2f b[4], q[4];
    sph4_primary_pk(S, d.x, d.y, d.z, b, q);
    if (ANY_LANE_RARE(any_nonneg(q[0], q[1], q[2], q[3]))) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t2;
          if (sph_exact(comp(b[i], c), comp(q[i], c), h[c].t, t2)) {
            h[c].t = t2;
            h[c].idx = idx + i;
          }
        }
    }
  };
  const int n8 = n & ~7;
  if (n8) {
    SphP2 A[4], B[4];
    fetch_batch(rec, 0, A);
    for (int k = 0; k < n8; k += 8) {
      fetch_batch(rec, rec.landed(A[3].zc, k + 4), B);
      test4(A, base + k);
      fetch_batch(rec, rec.landed(B[3].zc, min(k + 8, n - 4)), A);
      test4(B, base + k + 4);
    }
  }
}

// ---------------------------------------------------------------------------------------
// FILTERS: cheap conservative stand-ins for the reference's sphere discriminant.
//
// The reference arithmetic (no FMA, fixed order) costs 7 fp32 operations per
// primary (ray, sphere) pair after hoisting and 16 per shadow pair, and all of them but a handful
// per ray end in `disc < 0 -> miss`.  The hot loops therefore evaluate a cheaper expression q'
// with FUSED multiply-adds -- 4 operations per primary pair, 8 per shadow pair -- built so that
//
//        the reference's fp32 evaluation does not reject at `disc < 0`   ==>   q' >= 0,
//
// and run the reference arithmetic itself (unchanged, on the exact records) for every batch of 8
// spheres in which any lane has q' >= 0.  Accepts, t2 and the index order of ties are therefore
// exactly the reference's: q' decides only WHETHER the exact code runs, never a result.
// u = 2^-24, inputs finite with |coordinates| < 2^60, ray directions unit to a few ulp
// (they come out of normalize()).
//
// Primary pairs (oc hoisted, A = |ocx|+|ocy|+|ocz|):  reference b = fl-dot(oc,d) and filter
//   b' = fma(ocz,dz,fma(ocy,dy,fl(ocx dx))) are both within 3.01u A of the real dot product, so
//   b^2 <= b'^2 + 12.1u A^2; with bb = fl(b b) <= b^2 (1+u) + 2^-149 the reference's
//   "not disc < 0"  <=>  bb >= cc  implies  b'^2 - cc + 13.3u A^2 + 2^-149 >= 0.
//   With ccm = fl(cc - 2^-19 (A2f + r2) - 2^-120), A2f = fl(A)^2 >= A^2 (1-5u) -- the 32u margin
//   covers the 13.3u above, the rounding of ccm itself (<= u (A^2 + r2)) and the rounding of A2f --
//   this is  b'^2 >= ccm.  The kernel tests it in the scaled form |b''| >= 1 (3 operations per
//   pair + one v_max3 per sphere; sph4_primary_filter_pk below).
//
// Shadow pairs: rt_device.h DevSphPairF, proof next to sph_any_filter below.
// ---------------------------------------------------------------------------------------
struct SphF2 { // DevSphF seen as two aligned pairs: (sx, w), (sy, sz)
  v2f xw, yz;
};

// 4 spheres x 2 pixels: b''[i] = fma(sz, dz, fma(sy, dy, fma(sx, dx, w))); candidate iff |b''| >= 1.
// Scaling the hoisted oc by 1/s (rt_device.h DevSphF) turns "b'^2 >= ccm" into "|b''| >= 1": the
// discriminant's fourth FMA and the per-sphere constant in the comparison are gone.  |b'' - beta/s|
// <= 5.01u A / s (2u from the scaling, 3.01u from the chain; beta = the real dot product) and the
// unscaled chain b' of the proof above has |b' - beta| <= 3.01u A, so |b'| >= sqrt(ccm) implies
// |b''| >= (sqrt(ccm) - 8.02u A) / s >= 1 for the s of DevSphF.
DEVINL void sph4_primary_filter_pk(const SphF2 (&s)[4], v2f dx, v2f dy, v2f dz, v2f (&b)[4]) {
  asm("v_pk_fma_f32 %0, %[s0a], %[x], %[s0a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %[s1a], %[x], %[s1a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %[s2a], %[x], %[s2a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %[s3a], %[x], %[s3a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %[s0b], %[y], %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %[s1b], %[y], %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %[s2b], %[y], %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %[s3b], %[y], %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %[s0b], %[z], %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %1, %[s1b], %[z], %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %2, %[s2b], %[z], %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %3, %[s3b], %[z], %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "s_nop 0"
      : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
      : [x] "v"(dx), [y] "v"(dy), [z] "v"(dz), [s0a] "s"(s[0].xw), [s0b] "s"(s[0].yz),
        [s1a] "s"(s[1].xw), [s1b] "s"(s[1].yz), [s2a] "s"(s[2].xw), [s2b] "s"(s[2].yz),
        [s3a] "s"(s[3].xw), [s3b] "s"(s[3].yz));
}

// running max of |b''| over 4 spheres x 2 pixels (v_max3_f32 with |.| modifiers: one instruction
// per sphere)
DEVINL float max_abs8(const v2f (&b)[4], float m) {
  asm("v_max3_f32 %0, %0, |%1|, |%2|\n\t"
      "v_max3_f32 %0, %0, |%3|, |%4|\n\t"
      "v_max3_f32 %0, %0, |%5|, |%6|\n\t"
      "v_max3_f32 %0, %0, |%7|, |%8|"
      : "+v"(m)
      : "v"(b[0].x), "v"(b[0].y), "v"(b[1].x), "v"(b[1].y), "v"(b[2].x), "v"(b[2].y), "v"(b[3].x),
        "v"(b[3].y));
  return m;
}

// any q' >= 0 among 8 values?  raw-bit test: non-negative floats (and +NaN) are >= 0 as ints;
// a q' of -0 cannot occur (an exactly-zero fma result is +0 in round-to-nearest)
DEVINL int max_bits8(const v2f (&q)[4], int m) {
  m = max3i(m, __float_as_int(q[0].x), __float_as_int(q[0].y));
  m = max3i(m, __float_as_int(q[1].x), __float_as_int(q[1].y));
  m = max3i(m, __float_as_int(q[2].x), __float_as_int(q[2].y));
  return max3i(m, __float_as_int(q[3].x), __float_as_int(q[3].y));
}

// SMEM + 2 pixels per lane: the filter over 8 spheres per step; a step with a candidate re-reads
// its 8 EXACT records and runs the reference arithmetic on them (test_sph_primary, above).
// n is a multiple of 8 (the caller peels the rest through the exact loop).
template <typename FetchF, typename FetchE>
DEVINL void closest_sph_primary_filter(FetchF recf, FetchE rece, int n, int base, const V3<v2f> &d,
                                       Hit (&h)[2]) {
  auto test8 = [&](const SphF2(&S)[8], int k) {
    v2f q0[4], q1[4];
    const SphF2(&S0)[4] = reinterpret_cast<const SphF2(&)[4]>(S[0]);
    const SphF2(&S1)[4] = reinterpret_cast<const SphF2(&)[4]>(S[4]);
    sph4_primary_filter_pk(S0, d.x, d.y, d.z, q0);
    sph4_primary_filter_pk(S1, d.x, d.y, d.z, q1);
    const float m = max_abs8(q1, max_abs8(q0, 0.f));
    if (ANY_LANE_RARE(m >= 1.f)) {
      const V3<v2f> dv[1] = {d};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        DevSphP E[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) E[i] = rece(k + 4 * j + i);
        test_sph_primary<v2f, 1, 4>(E, base + k + 4 * j, dv, h);
      }
    }
  };
  if (n >= 8) {
    SphF2 A[8], B[8];
    fetch_batch(recf, 0, A);
    for (int k = 0; k < n; k += 16) {
      fetch_batch(recf, recf.landed(A[7].yz, min(k + 8, n - 8)), B);
      test8(A, k);
      if (k + 8 >= n) break; // odd number of 8-blocks: B was a clamped refetch, unused
      fetch_batch(recf, recf.landed(B[7].yz, min(k + 16, n - 8)), A);
      test8(B, k + 8);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Sphere GROUPS (rt_device.h SphGroups): the same scaled test on the bounding sphere (C, R) of
// kSphGroup spheres that are neighbours in space; only groups some ray of the wave may touch have
// their members filtered.  What R must hold -- member i's reference test does not reject at
// `disc < 0` for a ray (o, d), |d|^2 within 8u of 1; oc_i, b, cc_i the reference's fp32 values,
// beta_i = oc_i . d in real arithmetic, A_i = |oc_i|_1:
//   bb >= cc_i with |b - beta_i| <= 3.01u A_i and bb <= b^2 (1+u) + 2^-149 gives
//   cc_i <= beta_i^2 + 7.2u A_i^2 + 2^-149, and cc_i >= |oc_i|^2 - r_i^2 - 4.02u |oc_i|^2 - u r_i^2, so the
//   line's squared distance from the point c~_i = o - oc_i is
//   D_i^2 = |oc_i|^2 - beta_i^2 / |d|^2 <= r_i^2 (1+u) + 19.4u A_i^2 + 2^-149,
//   D_i <= r_i (1+u) + 4.41 sqrt(u) A_i + 2^-74.
//   c~_i is within 1.01u A_i of c_i (the rounding of oc_i), likewise C~ = o - OC of the stored
//   centre C (A_G = |OC|_1), and A_i <= (A_G + sqrt(3) |c_i - C|)(1 + 2u).  With rgeo >= r_i +
//   |c_i - C| for every member the line passes C~ within
//   rgeo (1+u) + (4.41 sqrt(u) + 2.1u)(A_G + 1.74 rgeo) + 2^-74  <=  R := rgeo + 0x1.2p-10 (A_G + 2 rgeo) + 2^-60
//   (4.41 * 2^-12 = 1.0767e-3 < 0x1.2p-10 = 1.0986e-3: 2 % to spare for evaluating R in fp32).
// "The line passes C~ within R" is |OC|^2 - beta_G^2 / |d|^2 <= R^2, i.e. beta_G^2 >= K (1 - 8u) with
// K = |OC|^2 - R^2 (K <= 0: the record is "always a candidate").  The record's ccm is at most
// K - 26.8u (A_G^2 + R2) (k_prepare_groups: R2 = R^2 * 1.00001, fp32 dot and subtraction 5.1u, margin
// 32u less the 5u of A2f) and the filter's b' has b'^2 >= beta_G^2 - 6.1u A_G^2 >= K - 14.1u A_G^2 >= ccm:
// the scaled test |b''| >= 1 follows exactly as for a single sphere (sph4_primary_filter_pk).
// A camera inside a member is inside (C, rgeo): K < 0, always a candidate.
//
// Tests happen in group order, not index order.  The reference keeps the FIRST of two equal
// closest t (strict `t2 < *t`): the sorted exact test below accepts an equal t iff its original
// index is lower, which is the same rule stated without reference to the order of the tests.
// ---------------------------------------------------------------------------------------
DEVINL void test_sph_primary_sorted(const DevSphP (&s)[4], const DevIdx4 &orig, int base,
                                    const V3<v2f> &d, Hit (&h)[2]) {
  v2f b[4], q[4];
  float m = -1.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b[i] = (s[i].ocx * d.x + s[i].ocy * d.y) + s[i].ocz * d.z;
    q[i] = b[i] * b[i] - s[i].cc;
    m = fmaxf(m, fmaxf(q[i].x, q[i].y));
  }
  if (ANY_LANE_RARE(!(m < 0.f))) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t2;
        if (sph_exact_nb(comp(b[i], c), comp(q[i], c), t2)) {
          const int id = base + orig.v[i];
          if (t2 < h[c].t || (t2 == h[c].t && id < h[c].idx)) {
            h[c].t = t2;
            h[c].idx = id;
          }
        }
      }
  }
}

// max / min of the two halves' magnitudes in ONE instruction (|.| source modifiers)
DEVINL float max_abs2(v2f q) {
  float t;
  asm("v_max3_f32 %0, |%1|, |%2|, |%2|" : "=v"(t) : "v"(q.x), "v"(q.y));
  return t;
}
DEVINL float min_abs2(v2f q) {
  float t;
  asm("v_min3_f32 %0, |%1|, |%2|, |%2|" : "=v"(t) : "v"(q.x), "v"(q.y));
  return t;
}

// the scaled test on 8 records x 2 pixels: wave-uniform mask of the records SOME lane passes
// (0 for nearly every step: the per-record ballots sit behind one wave-wide check)
DEVINL uint32_t sph8_primary_mask(const SphF2 (&S)[8], const V3<v2f> &d) {
  v2f q0[4], q1[4];
  const SphF2(&S0)[4] = reinterpret_cast<const SphF2(&)[4]>(S[0]);
  const SphF2(&S1)[4] = reinterpret_cast<const SphF2(&)[4]>(S[4]);
  sph4_primary_filter_pk(S0, d.x, d.y, d.z, q0);
  sph4_primary_filter_pk(S1, d.x, d.y, d.z, q1);
  const float m = max_abs8(q1, max_abs8(q0, 0.f));
  uint32_t mask = 0;
  if (ANY_LANE_RARE(m >= 1.f)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const v2f q = j < 4 ? q0[j & 3] : q1[j & 3];
      if (__builtin_amdgcn_ballot_w64(max_abs2(q) >= 1.f) != 0) mask |= 1u << j;
    }
  }
  return mask;
}

// Three-level sweep (rt_device.h SphGroups): 8 hyper-groups per step, double-buffered; an opened
// hyper-group costs one step over its 8 super-groups, an opened super-group one over its 8 groups,
// an opened group one over its 8 spheres' filter records, and only spheres that pass that run the
// reference arithmetic.  Hyper-group y holds super-groups [8 y, 8 y + 8), super-group s groups
// [8 s, 8 s + 8), group g the sorted slots [8 g, 8 g + 8); n_hyp is a multiple of 8 (pad records
// never pass).
// the 8 member spheres of leaf group g (sorted slots [8 g, 8 g + 8)): filter, then the reference
// arithmetic on the flagged halves
template <typename FetchF, typename FetchE, typename FetchI>
DEVINL void sph_group_members_primary(FetchF recf, FetchE rece, FetchI reci, int g, int base,
                                      const V3<v2f> &d, Hit (&h)[2]) {
  SphF2 S[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) S[i] = recf(8 * g + i);
  const uint32_t mask = sph8_primary_mask(S, d);
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (mask & (0xFu << (4 * j))) {
      DevSphP E[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) E[i] = rece(8 * g + 4 * j + i);
      test_sph_primary_sorted(E, reci(2 * g + j), base, d, h);
    }
}

template <typename FetchF, typename FetchE, typename FetchI>
DEVINL void closest_sph_primary_groups(FetchF recy, FetchF recu, FetchF recg, FetchF recf, FetchE rece,
                                       FetchI reci, int n_hyp, int base, const V3<v2f> &d,
                                       Hit (&h)[2]) {
  static_assert(kSphGroup == 8 && kSphSuper == 8 && kSphHyper == 8 && kSphGroupStep == 8,
                "8-wide bodies below");
  auto members = [&](int g) { sph_group_members_primary(recf, rece, reci, g, base, d, h); };
  auto groups = [&](int s) {
    SphF2 G[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) G[i] = recg(8 * s + i);
    uint32_t mask = sph8_primary_mask(G, d);
    while (mask) {
      const int j = __builtin_ctz(mask);
      mask &= mask - 1;
      members(8 * s + j);
    }
  };
  auto supers = [&](int y) { // the 8 super-groups of hyper-group y
    SphF2 U[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) U[i] = recu(8 * y + i);
    uint32_t mask = sph8_primary_mask(U, d);
    while (mask) {
      const int j = __builtin_ctz(mask);
      mask &= mask - 1;
      groups(8 * y + j);
    }
  };
  auto step = [&](const SphF2(&Y)[8], int y0) {
    uint32_t mask = sph8_primary_mask(Y, d);
    while (mask) {
      const int j = __builtin_ctz(mask);
      mask &= mask - 1;
      supers(y0 + j);
    }
  };
  SphF2 A[8], B[8];
  fetch_batch(recy, 0, A);
  for (int y = 0; y < n_hyp; y += 16) {
    fetch_batch(recy, recy.landed(A[7].yz, min(y + 8, n_hyp - 8)), B);
    step(A, y);
    if (y + 8 >= n_hyp) break; // odd number of steps: B was a clamped refetch, unused
    fetch_batch(recy, recy.landed(B[7].yz, min(y + 16, n_hyp - 8)), A);
    step(B, y + 8);
  }
}

// ---------------------------------------------------------------------------------------
// Triangle FILTERS.  ray_triangle.h's accept needs (with s = sign(det), all in its own rounded
// arithmetic):  s*un > 0,  s*vn > 0,  s*(un + vn) <= |det| (1 + 4u)   -- from u2 >= eps, v2 >= eps,
// u2 + v2 <= 1 -- and |un|, |vn| <= |det| (1 + 4u).  In product form, free of sign logic:
//     un*det > 0,   vn*det > 0,   det*(det - un - vn) >= -4u det^2.
// The filter evaluates det', un', vn' as FMA dot products with hoisted cross products (rt_device.h
// DevTriF / DevTriPairF) and passes the pair on when
//     A = un'*det' + M >= 0,   B = vn'*det' + M >= 0,   C = det'*(det' - (un' + vn')) + M >= 0
// (each the single rounding of an exact expression: sign-exact).  With Ed, Eu, Ev the distances
// between the filter's and the reference's numerators and Dmax >= |det_ref|:
//     un'*det' >= un*det - (|un| Ed + |det| Eu + Eu Ed) >= -(Dmax (Ed (1+4u) + Eu) + Eu Ed)
// and likewise for B; for C, |det - un - vn| <= 3 Dmax under the accept hypothesis and the filter's
// w = det' - (un' + vn') is off by Ew = Ed + Eu + Ev + 2u (|det'| + |un'| + |vn'|):
//     C' >= -(Dmax (4u Dmax + Ew + 3 Ed) + Ed Ew).
// Primary rays, 1-norms P12 = |e1||e2|, Pt2 = |tv||e2|, Q = |qv|:  cross products carry 2.01u of
// their products' magnitudes, an FMA dot product 3.01u, the reference's det / un  10.04u P12 / Pt2:
// Ed <= 16u P12, Eu <= 16u Pt2, Ev <= 8u Q, Dmax = 1.0001 P12, so every margin above is below
// 70.1u P12 (P12 + Pt2 + Q); M = 2^-17 P12 (P12 + Pt2 + Q) = 128u P12 (...).
// Shadow rays (origin o, a = fl(o - g), m = fl(a x L), v0' = fl(v0 - g), R = |a| + |v0'|):
// Eu <= 22u |e2| R, Ev <= 22u |e1| R (the reference's own 10.04u |tv||e2|, the roundings of a, v0'
// and m, and the six-term FMA chain), so the margins are below u P12 (70.1 P12 + 24.1 (|e1|+|e2|) R);
// M = 128u P12 (P12 + (|e1|+|e2|) (|v0'| + rho_max)) covers every ray with |a| <= rho_max, and a
// ray that starts further out is sent to the exact code for every triangle.
// ---------------------------------------------------------------------------------------
struct TriF { // DevTriF as six aligned pairs
  v2f a, b, c, d, e, pad; // (n1x,n1y) (n1z,n2x) (n2y,n2z) (n3x,n3y) (n3z,M) (-,-)
};

// 2 triangles x 2 pixels -> A, B, C of each (candidate iff all three >= 0 for a pixel)
DEVINL void tri2_primary_filter_pk(const TriF (&T)[2], v2f dx, v2f dy, v2f dz, v2f (&A)[2],
                                   v2f (&B)[2], v2f (&C)[2]) {
  v2f det0, det1, s0, s1;
  asm("v_pk_mul_f32 %[det0], %[t0a], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[det1], %[t1a], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[un0], %[t0b], %[x] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[un1], %[t1b], %[x] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[vn0], %[t0d], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[vn1], %[t1d], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %[det0], %[t0a], %[y], %[det0] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[det1], %[t1a], %[y], %[det1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[un0], %[t0c], %[y], %[un0] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[un1], %[t1c], %[y], %[un1] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[vn0], %[t0d], %[y], %[vn0] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[vn1], %[t1d], %[y], %[vn1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[det0], %[t0b], %[z], %[det0] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[det1], %[t1b], %[z], %[det1] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[un0], %[t0c], %[z], %[un0] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[un1], %[t1c], %[z], %[un1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[vn0], %[t0e], %[z], %[vn0] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %[vn1], %[t1e], %[z], %[vn1] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_add_f32 %[s0], %[un0], %[vn0]\n\t"
      "v_pk_add_f32 %[s1], %[un1], %[vn1]\n\t"
      "v_pk_fma_f32 %[un0], %[un0], %[det0], %[t0e] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[un1], %[un1], %[det1], %[t1e] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n\t"
      "v_pk_add_f32 %[s0], %[det0], %[s0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[s1], %[det1], %[s1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %[vn0], %[vn0], %[det0], %[t0e] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[vn1], %[vn1], %[det1], %[t1e] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[s0], %[det0], %[s0], %[t0e] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[s1], %[det1], %[s1], %[t1e] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n\t"
      "s_nop 0"
      : [un0] "=&v"(A[0]), [un1] "=&v"(A[1]), [vn0] "=&v"(B[0]), [vn1] "=&v"(B[1]), [s0] "=&v"(C[0]),
        [s1] "=&v"(C[1]), [det0] "=&v"(det0), [det1] "=&v"(det1)
      : [x] "v"(dx), [y] "v"(dy), [z] "v"(dz), [t0a] "s"(T[0].a), [t0b] "s"(T[0].b),
        [t0c] "s"(T[0].c), [t0d] "s"(T[0].d), [t0e] "s"(T[0].e), [t1a] "s"(T[1].a),
        [t1b] "s"(T[1].b), [t1c] "s"(T[1].c), [t1d] "s"(T[1].d), [t1e] "s"(T[1].e));
  (void)s0;
  (void)s1;
}

// sign test of (A, B, C): a pixel is a candidate iff all three are >= 0, i.e. iff the OR of their
// bit patterns has a clear sign bit (-0 cannot occur: an exact-zero fma result is +0)
DEVINL int tri_flags(v2f A, v2f B, v2f C, int m) {
  const int ox = __float_as_int(A.x) | __float_as_int(B.x) | __float_as_int(C.x);
  const int oy = __float_as_int(A.y) | __float_as_int(B.y) | __float_as_int(C.y);
  return max3i(m, ox, oy);
}

// ---------------------------------------------------------------------------------------
// Triangle pre-filter.  With o' = v0 + tvec (the origin the reference's numerators are taken from)
// the line o' + t d meets the triangle's plane at X* = v0 + u* e1 + v* e2, u* = un*/det*, v* =
// vn*/det* (exact Cramer).  The reference's fp32 numerators are within eu = 10.04u |tv||e2|,
// ev = 5.04u |tv||e1|, ed = 10.05u |e1||e2| (1-norms) of un*, vn*, det*, and an accept needs
// 0 < un/det <= 1+4u, 0 < vn/det, (un+vn)/det <= 1+4u; hence u* >= -Du, v* >= -Dv,
// u* + v* <= 1 + 4u + Du + Dv with Du + Dv <= (eu + ev + 2.0001 ed) / |det*|, and X* lies within
// (3 (Du + Dv) + 4u) emax of the triangle (clamp the negatives, rescale the sum).  So for
//     |det*| >= tau := 3.2u (10.04 |tv||e2| + 5.04 |tv||e1| + 20.1 |e1||e2|) emax / rho
// X* is within rho (the bounding radius around the centroid G) of the triangle, i.e. the line
// passes within R = 2 rho of G; and |det*| < tau shows as |det'| <= tau' = tau + 10.1u |e1||e2| in
// the filter's own FMA dot product.  A reference accept therefore raises at least one of
//     |b''| >= 1   (sphere (G, R) in the scaled form of DevSphF, margins as for spheres)
//     |g''| <= 1   (g'' = d . n1 / tau')
// and only then does the triangle filter above (and, behind it, the reference arithmetic) run.
// Slivers (rho < 2^-10 emax) are always passed on.  6 packed FMAs + 2 three-input min/max per
// triangle per 128 rays instead of 14 + 3.
// ---------------------------------------------------------------------------------------
struct TriPF { // DevTriPF as four aligned pairs
  v2f xw, yz, gxy, gz;
};

// 4 triangles x 2 pixels -> b''[i], g''[i]
DEVINL void tri4_primary_prefilter_pk(const TriPF (&T)[4], v2f dx, v2f dy, v2f dz, v2f (&b)[4],
                                      v2f (&g)[4]) {
  asm("v_pk_fma_f32 %0, %[t0a], %[x], %[t0a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %[t1a], %[x], %[t1a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %[t2a], %[x], %[t2a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %[t3a], %[x], %[t3a] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"
      "v_pk_mul_f32 %4, %[t0c], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %5, %[t1c], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %6, %[t2c], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %7, %[t3c], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %[t0b], %[y], %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %[t1b], %[y], %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %2, %[t2b], %[y], %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %3, %[t3b], %[y], %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %4, %[t0c], %[y], %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %5, %[t1c], %[y], %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %6, %[t2c], %[y], %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %7, %[t3c], %[y], %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %0, %[t0b], %[z], %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %1, %[t1b], %[z], %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %2, %[t2b], %[z], %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %3, %[t3b], %[z], %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %4, %[t0d], %[z], %4 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %5, %[t1d], %[z], %5 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %6, %[t2d], %[z], %6 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %7, %[t3d], %[z], %7 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
      "s_nop 0"
      : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]),
        "=&v"(g[3])
      : [x] "v"(dx), [y] "v"(dy), [z] "v"(dz), [t0a] "s"(T[0].xw), [t0b] "s"(T[0].yz),
        [t0c] "s"(T[0].gxy), [t0d] "s"(T[0].gz), [t1a] "s"(T[1].xw), [t1b] "s"(T[1].yz),
        [t1c] "s"(T[1].gxy), [t1d] "s"(T[1].gz), [t2a] "s"(T[2].xw), [t2b] "s"(T[2].yz),
        [t2c] "s"(T[2].gxy), [t2d] "s"(T[2].gz), [t3a] "s"(T[3].xw), [t3b] "s"(T[3].yz),
        [t3c] "s"(T[3].gxy), [t3d] "s"(T[3].gz));
}

// max |b''| and min |g''| over two records x 2 pixels (one 3-input instruction per record each)
DEVINL void maxmin_abs4(v2f b0, v2f b1, v2f g0, v2f g1, float &mx, float &mn) {
  asm("v_max3_f32 %0, %0, |%2|, |%3|\n\t"
      "v_min3_f32 %1, %1, |%6|, |%7|\n\t"
      "v_max3_f32 %0, %0, |%4|, |%5|\n\t"
      "v_min3_f32 %1, %1, |%8|, |%9|"
      : "+v"(mx), "+v"(mn)
      : "v"(b0.x), "v"(b0.y), "v"(b1.x), "v"(b1.y), "v"(g0.x), "v"(g0.y), "v"(g1.x), "v"(g1.y));
}

// SMEM + 2 pixels per lane: pre-filter over 4 triangles per step (double-buffered scalar fetches);
// a flagged pair of triangles goes through the triangle filter, and what that flags through the
// reference arithmetic (test_tri2_primary on the exact DevTriP records).  n is a multiple of 4.
template <typename FetchP, typename FetchF, typename FetchE>
DEVINL void closest_tri_primary_filter(FetchP recp, FetchF recf, FetchE rece, int n, int base,
                                       const V3<v2f> &d, Hit (&h)[2]) {
  auto level2 = [&](int k) { // triangles k, k+1
    const TriF T[2] = {recf(k), recf(k + 1)};
    v2f A[2], B[2], C[2];
    tri2_primary_filter_pk(T, d.x, d.y, d.z, A, B, C);
    const int m = tri_flags(A[1], B[1], C[1], tri_flags(A[0], B[0], C[0], -1));
    if (ANY_LANE_RARE(m >= 0)) {
      const V3<v2f> dv[1] = {d};
      const DevTriP E[2] = {rece(k), rece(k + 1)};
      test_tri2_primary<v2f, 1>(E, base + k, dv, h);
    }
  };
  auto test4 = [&](const TriPF(&T)[4], int k) {
    v2f b[4], g[4];
    tri4_primary_prefilter_pk(T, d.x, d.y, d.z, b, g);
    float mx0 = 0.f, mn0 = 2.f, mx1 = 0.f, mn1 = 2.f;
    maxmin_abs4(b[0], b[1], g[0], g[1], mx0, mn0);
    maxmin_abs4(b[2], b[3], g[2], g[3], mx1, mn1);
    const bool f0 = (mx0 >= 1.f) | (mn0 <= 1.f), f1 = (mx1 >= 1.f) | (mn1 <= 1.f);
    if (ANY_LANE_RARE(f0 | f1)) {
      if (__builtin_amdgcn_ballot_w64(f0)) level2(k);
      if (__builtin_amdgcn_ballot_w64(f1)) level2(k + 2);
    }
  };
  if (n >= 4) {
    TriPF A[4], B[4];
    fetch_batch(recp, 0, A);
    for (int k = 0; k < n; k += 8) {
      fetch_batch(recp, recp.landed(A[3].gz, min(k + 4, n - 4)), B);
      test4(A, k);
      if (k + 4 >= n) break;
      fetch_batch(recp, recp.landed(B[3].gz, min(k + 8, n - 4)), A);
      test4(B, k + 4);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Triangle GROUPS (rt_device.h TriGroups): the pre-filter's two statements, lifted from one
// triangle to 8 (a group), 128 (a super-group) and 1,024 (a hyper-group) that are neighbours in
// space.  For a member t the pre-filter's proof gives: a reference accept has (S_t) the line
// through o'_t = v0 + tvec within rho_t of a point of the triangle, or (E_t) |det*| < tau_t.
//  (S_t): o'_t is within 1.01u |tvec_t|_1 of the camera o, C~ = o - fl(o - C) within 1.01u A of the
//   stored centre C (A = |fl(o - C)|_1), |tvec_t|_1 <= (A + rext)(1 + 2u); with rgeo >= rho_t +
//   |v - C| for every vertex v of every member (a triangle is the convex hull of its vertices)
//   the line through o passes C~ within  R := rgeo + 8u (A + rext) + 2^-60, and (C, R) goes
//   through the scaled sphere record exactly as a sphere group's (C, R) does: |b''| >= 1.
//  (E_t): |d . n_t| < tau_t / |n1_t| <= b0 + b1 |tvec_t|_1 for the unit normal n_t (tau_t's formula is
//   linear in |tvec|: host, tri_group_bounds).  With a unit axis a and sin(angle(a, +-n_t)) <= smax:
//   |d . a| <= |d . n_t| + |d| smax.  The record holds a / kappa', kappa' = (smax + b0 + b1 (A + rext)
//   + 2^-20) * 1.0001, so |g''| <= 1 for the FMA chain (its own 3.01u |a|_1 / kappa' sits inside
//   the 2^-20).  kappa' >= 1 (no useful cone), a sliver among the members: the group is always open.
//  (P) Primary rays only: (E_t) can carry an accept only when the camera is nearly IN the member's
//   plane.  With tvec = p + h n_t (p in the plane), un* = tvec . (d x e2) = lambda (d . n_t) + h d . (e2 x n_t)
//   and vn* = d . (tvec x e1) = mu (d . n_t) + h d . (n_t x e1), |lambda| <= |e2||tvec|, |mu| <= |e1||tvec|.
//   An accept has |un|, |vn| <= |det| (1 + 4u), so with the reference's error bounds eu, ev, ed and
//   |det*| < tau:  |un*| < U := (tau + ed)(1 + 4u) + eu,  |vn*| < V := (tau + ed)(1 + 4u) + ev, hence
//   |h| |d . w2| <= U / |e2| + |tvec| tau / |n1|  and  |h| |d . w1| <= V / |e1| + |tvec| tau / |n1|  for the
//   in-plane unit vectors w1, w2 across e1, e2.  They enclose the triangle's angle phi, so one of
//   |d . w_i| is at least |d_par| min(sin, cos)(phi / 2) >= 0.99 gamma, gamma = |n1| / (2 |e1||e2|), as long
//   as |d . n_t| < 0.1:   |h| <= H_t := [max(V / |e1|, U / |e2|) + |tvec| tau / |n1|] / (0.99 gamma),
//   a few hundredths of a unit on c5.  The camera is one point per frame: k_prepare_tri_groups
//   evaluates |h| <= H_t for every member of every group and super-group (tri_escape_possible), and
//   where no member passes, no ray of the frame can be accepted through (E_t) by any of them:
//   the record's cone part is switched off.  Only groups with a member seen edge-on keep it.
//  (K) The two halves trade against each other, level by level.  The pre-filter's proof puts an
//   accepted hit point within (3 (Du + Dv) + 4u) emax of its triangle with Du + Dv <= (eu + ev +
//   2.0001 ed) / |det*|; tau_t makes that <= rho_t for |det*| >= tau_t.  For |det*| >= tau_t / k it is
//   <= 0.9375 k rho_t + 4u emax <= k rho_t (rho_t > 2^-10 emax, k >= 1).  So a level may state
//   (S_t) with k rho_t (rgeo >= k rho_t + |v - C|) and (E_t), (P) with tau_t / k -- each level
//   independently, because an accept implies every level's own "(S) or (E)".  Groups use k = 1;
//   a super- or hyper-group, whose sphere is large anyway, takes the k that doubles its tight
//   radius (k rho_max <= max |v - C|; host, tri_group_bounds), at most kTriSlackSuper / kTriSlackHyper:
//   its "nearly parallel" band is k times thinner, H_t shrinks with tau, and a ray outside an
//   upper level's band and sphere never opens the chain below (c5 k_primary 7.9 -> 4.7 ms).  The
//   frame's cones come in three chains accordingly: a group's members are evaluated at k = 1, at
//   its super-group's k and at its hyper-group's k (k_prepare_tri_groups / _merge).  Shadow
//   records keep k = 1: their static cones are dominated by the sine term.
// A member accept therefore opens its group and super-group; inside an opened group the per-triangle
// pre-filter, the filter and the reference arithmetic run as before.  Order: as for the sphere
// groups, an equal closest t goes to the lower ORIGINAL index.
// ---------------------------------------------------------------------------------------
DEVINL void test_tri2_primary_sorted(const DevTriP (&T)[2], int id0, int id1, const V3<v2f> &d,
                                     Hit (&h)[2]) {
  v2f det[2], un[2], vn[2];
  bool any = false;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const V3<v2f> pv = cross_vu(d, ld3(T[i].e2)); // ray_triangle.h:18
    det[i] = dotu(ld3(T[i].e1), pv);              // :21
    un[i] = dotu(ld3(T[i].tv), pv);               // :32 numerator
    vn[i] = dotu(ld3(T[i].qv), d);                // :40 numerator
#pragma unroll
    for (int c = 0; c < 2; ++c)
      any |= tri_candidate(comp(det[i], c), comp(un[i], c), comp(vn[i], c));
  }
  if (ANY_LANE_RARE(any)) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float de = comp(det[i], c), u = comp(un[i], c), v = comp(vn[i], c);
        float t2, v2;
        if (tri_candidate(de, u, v) && tri_exact_nb(de, u, v, T[i].tnum, t2, v2)) {
          const int id = i ? id1 : id0;
          if (t2 < h[c].t || (t2 == h[c].t && id < h[c].idx)) {
            h[c].t = t2;
            h[c].v = v2;
            h[c].idx = id;
          }
        }
      }
  }
}

// n_hyp is a multiple of kTriGroupStep (= 4; pad records never open).  Hyper-group y holds
// super-groups [kTriHyper y, kTriHyper (y + 1)), super-group s groups [kTriSuper s, kTriSuper (s + 1)),
// group g the sorted slots [8 g, 8 g + 8).
// pre-filter body on 4 records x 2 pixels: per-lane flags of the two record pairs
DEVINL void tri4_primary_flags(const TriPF (&T)[4], const V3<v2f> &d, v2f (&b)[4], v2f (&g)[4], bool &f0,
                               bool &f1) {
  tri4_primary_prefilter_pk(T, d.x, d.y, d.z, b, g);
  float mx0 = 0.f, mn0 = 2.f, mx1 = 0.f, mn1 = 2.f;
  maxmin_abs4(b[0], b[1], g[0], g[1], mx0, mn0);
  maxmin_abs4(b[2], b[3], g[2], g[3], mx1, mn1);
  f0 = (mx0 >= 1.f) | (mn0 <= 1.f);
  f1 = (mx1 >= 1.f) | (mn1 <= 1.f);
}
// sorted triangles k .. k+3 (k a multiple of 4): pre-filter; flagged pairs: filter; flagged again:
// the reference arithmetic, an equal closest t going to the lower ORIGINAL index
template <typename FetchP, typename FetchF, typename FetchE, typename FetchI>
DEVINL void tri_members4_primary(FetchP recp, FetchF recf, FetchE rece, FetchI reci, int k,
                                 const V3<v2f> &d, Hit (&h)[2]) {
  auto level2 = [&](int k2) { // sorted triangles k2, k2+1 (k2 even)
    const TriF T[2] = {recf(k2), recf(k2 + 1)};
    v2f A[2], B[2], C[2];
    tri2_primary_filter_pk(T, d.x, d.y, d.z, A, B, C);
    const int m = tri_flags(A[1], B[1], C[1], tri_flags(A[0], B[0], C[0], -1));
    if (ANY_LANE_RARE(m >= 0)) {
      const DevTriP E[2] = {rece(k2), rece(k2 + 1)};
      const DevIdx4 I = reci(k2 >> 2);
      test_tri2_primary_sorted(E, (k2 & 2) ? I.v[2] : I.v[0], (k2 & 2) ? I.v[3] : I.v[1], d, h);
    }
  };
  TriPF T[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) T[i] = recp(k + i);
  v2f b[4], g[4];
  bool f0, f1;
  tri4_primary_flags(T, d, b, g, f0, f1);
  if (ANY_LANE_RARE(f0 | f1)) {
    if (__builtin_amdgcn_ballot_w64(f0)) level2(k);
    if (__builtin_amdgcn_ballot_w64(f1)) level2(k + 2);
  }
}

template <typename FetchP, typename FetchF, typename FetchE, typename FetchI>
DEVINL void closest_tri_primary_groups(FetchP recy, FetchP recu, FetchP recg, FetchP recp, FetchF recf,
                                       FetchE rece, FetchI reci, int n_hyp, const V3<v2f> &d,
                                       Hit (&h)[2]) {
  static_assert(kTriGroup == 8 && kTriSuper % 4 == 0 && kTriHyper % 4 == 0 && kTriGroupStep == 4,
                "4-wide bodies below");
  auto flags4 = [&](const TriPF(&T)[4], v2f (&b)[4], v2f (&g)[4], bool &f0, bool &f1) {
    tri4_primary_flags(T, d, b, g, f0, f1);
  };
  auto members4 = [&](int k) { tri_members4_primary(recp, recf, rece, reci, k, d, h); };
  // wave-uniform mask of the records (groups / super-groups) some lane must open
  auto open4 = [&](const TriPF(&T)[4]) -> uint32_t {
    v2f b[4], g[4];
    bool f0, f1;
    flags4(T, b, g, f0, f1);
    uint32_t mask = 0;
    if (ANY_LANE_RARE(f0 | f1)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // two ballots OR-ed on the scalar side (a per-lane `||` costs more vector instructions)
        if ((__builtin_amdgcn_ballot_w64(max_abs2(b[j]) >= 1.f) |
             __builtin_amdgcn_ballot_w64(min_abs2(g[j]) <= 1.f)) != 0)
          mask |= 1u << j;
      }
    }
    return mask;
  };
  auto groups = [&](int s) { // the kTriSuper groups of super-group s, 4 at a time
    for (int q4 = 0; q4 < kTriSuper; q4 += 4) {
      TriPF G[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) G[i] = recg(kTriSuper * s + q4 + i);
      uint32_t mask = open4(G);
      while (mask) {
        const int j = __builtin_ctz(mask);
        mask &= mask - 1;
        const int g0 = kTriGroup * (kTriSuper * s + q4 + j);
        members4(g0);
        members4(g0 + 4);
      }
    }
  };
  auto supers = [&](int y) { // the kTriHyper super-groups of hyper-group y, 4 at a time
    for (int q4 = 0; q4 < kTriHyper; q4 += 4) {
      TriPF U[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) U[i] = recu(kTriHyper * y + q4 + i);
      uint32_t mask = open4(U);
      while (mask) {
        const int j = __builtin_ctz(mask);
        mask &= mask - 1;
        groups(kTriHyper * y + q4 + j);
      }
    }
  };
  auto step = [&](const TriPF(&Y)[4], int y0) {
    uint32_t mask = open4(Y);
    while (mask) {
      const int j = __builtin_ctz(mask);
      mask &= mask - 1;
      supers(y0 + j);
    }
  };
  TriPF A[4], B[4];
  fetch_batch(recy, 0, A);
  for (int y = 0; y < n_hyp; y += 8) {
    fetch_batch(recy, recy.landed(A[3].gz, min(y + 4, n_hyp - 4)), B);
    step(A, y);
    if (y + 4 >= n_hyp) break;
    fetch_batch(recy, recy.landed(B[3].gz, min(y + 8, n_hyp - 4)), A);
    step(B, y + 4);
  }
}

struct TriPairF { // DevTriPairF as sixteen aligned pairs
  v2f n1x, n1y, n1z, e1x, e1y, e1z, e2x, e2y, e2z, k1x, k1y, k1z, k2x, k2y, k2z, M;
};
struct RayTF { // one shadow ray in triangle-filter form: L and m = (o - g) x L as register pairs
  v2f Lxy;  // (Lx, Ly)
  v2f Lz_m; // (Lz, mx)
  v2f myz;  // (my, mz)
};
DEVINL RayTF make_ray_tri_filter(f3 o, f3 L, const float (&g)[3], float rho_max, bool &far) {
  const float ax = o.x - g[0], ay = o.y - g[1], az = o.z - g[2];
  far = !((fabsf(ax) + fabsf(ay)) + fabsf(az) <= rho_max); // also catches NaN
  RayTF r;
  r.Lxy = v2f{L.x, L.y};
  r.Lz_m = v2f{L.z, ay * L.z - az * L.y};
  r.myz = v2f{az * L.x - ax * L.z, ax * L.y - ay * L.x};
  return r;
}

// 2 pair records (4 triangles) x 1 ray -> A, B, C per record (halves = the record's two triangles)
DEVINL void tripair2_any_filter_pk(const TriPairF (&R)[2], const RayTF &r, v2f (&A)[2], v2f (&B)[2],
                                   v2f (&C)[2]) {
  v2f det0, det1;
  asm("v_pk_mul_f32 %[det0], %[r0n1x], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[x0], %[r0k2x], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[y0], %[r0k1x], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[det1], %[r1n1x], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[x1], %[r1k2x], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[y1], %[r1k1x], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %[det0], %[r0n1y], %[lxy], %[det0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0k2y], %[lxy], %[x0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0k1y], %[lxy], %[y0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[det1], %[r1n1y], %[lxy], %[det1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1k2y], %[lxy], %[x1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1k1y], %[lxy], %[y1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[det0], %[r0n1z], %[lzm], %[det0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0k2z], %[lzm], %[x0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0k1z], %[lzm], %[y0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[det1], %[r1n1z], %[lzm], %[det1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1k2z], %[lzm], %[x1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1k1z], %[lzm], %[y1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      // un = e2.m - L.k2 (into x), vn = L.k1 - e1.m (into y)
      "v_pk_fma_f32 %[x0], %[r0e2x], %[lzm], %[x0] op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0e1x], %[lzm], %[y0] op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %[x1], %[r1e2x], %[lzm], %[x1] op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1e1x], %[lzm], %[y1] op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %[x0], %[r0e2y], %[myz], %[x0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0e1y], %[myz], %[y0] op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %[x1], %[r1e2y], %[myz], %[x1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1e1y], %[myz], %[y1] op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %[x0], %[r0e2z], %[myz], %[x0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0e1z], %[myz], %[y0] op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %[x1], %[r1e2z], %[myz], %[x1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1e1z], %[myz], %[y1] op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
      "v_pk_add_f32 %[s0], %[x0], %[y0]\n\t"
      "v_pk_add_f32 %[s1], %[x1], %[y1]\n\t"
      "v_pk_fma_f32 %[x0], %[x0], %[det0], %[r0M]\n\t"
      "v_pk_fma_f32 %[x1], %[x1], %[det1], %[r1M]\n\t"
      "v_pk_add_f32 %[s0], %[det0], %[s0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[s1], %[det1], %[s1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %[y0], %[y0], %[det0], %[r0M]\n\t"
      "v_pk_fma_f32 %[y1], %[y1], %[det1], %[r1M]\n\t"
      "v_pk_fma_f32 %[s0], %[det0], %[s0], %[r0M]\n\t"
      "v_pk_fma_f32 %[s1], %[det1], %[s1], %[r1M]\n\t"
      "s_nop 0"
      : [x0] "=&v"(A[0]), [x1] "=&v"(A[1]), [y0] "=&v"(B[0]), [y1] "=&v"(B[1]), [s0] "=&v"(C[0]),
        [s1] "=&v"(C[1]), [det0] "=&v"(det0), [det1] "=&v"(det1)
      : [lxy] "v"(r.Lxy), [lzm] "v"(r.Lz_m), [myz] "v"(r.myz), [r0n1x] "s"(R[0].n1x),
        [r0n1y] "s"(R[0].n1y), [r0n1z] "s"(R[0].n1z), [r0e1x] "s"(R[0].e1x), [r0e1y] "s"(R[0].e1y),
        [r0e1z] "s"(R[0].e1z), [r0e2x] "s"(R[0].e2x), [r0e2y] "s"(R[0].e2y), [r0e2z] "s"(R[0].e2z),
        [r0k1x] "s"(R[0].k1x), [r0k1y] "s"(R[0].k1y), [r0k1z] "s"(R[0].k1z), [r0k2x] "s"(R[0].k2x),
        [r0k2y] "s"(R[0].k2y), [r0k2z] "s"(R[0].k2z), [r0M] "s"(R[0].M), [r1n1x] "s"(R[1].n1x),
        [r1n1y] "s"(R[1].n1y), [r1n1z] "s"(R[1].n1z), [r1e1x] "s"(R[1].e1x), [r1e1y] "s"(R[1].e1y),
        [r1e1z] "s"(R[1].e1z), [r1e2x] "s"(R[1].e2x), [r1e2y] "s"(R[1].e2y), [r1e2z] "s"(R[1].e2z),
        [r1k1x] "s"(R[1].k1x), [r1k1y] "s"(R[1].k1y), [r1k1z] "s"(R[1].k1z), [r1k2x] "s"(R[1].k2x),
        [r1k2y] "s"(R[1].k2y), [r1k2z] "s"(R[1].k2z), [r1M] "s"(R[1].M));
}

// ---- any-hit (main.cpp:314-329), general origin -------------------------------------------
// Per-pixel state of one occlusion() call.  tb is the bound: > 0 while the ray is still
// looking, set to 0 once it found its FIRST occluder (or if it never looked), so later
// primitives cannot accept (accepts need eps <= t2 < tb).  tocc receives that occluder's t2
// (occlusion() mutates the caller's t, quirk S3) and kocc its index in (triangles, spheres)
// order.  The wave leaves a loop early once no ray is looking (checked per block of
// kExitStride primitives, not per primitive).
struct Any {
  float tb;
  float tocc;
  int32_t kocc;
  // Group sweeps of a light that is NOT the last one (rt_device.h SphGroups / TriGroups): the next
  // light's ray starts at the FIRST occluder in index order (quirk S3), and the groups are not
  // swept in index order -- so such a sweep visits every group, keeps the ray looking, and
  // records the accepted primitive with the LOWEST original index: `orig` (wave-uniform, else
  // nullptr) maps a sorted slot, idx - orig_bias, to it; orig_add is n_tri for spheres.
  const int32_t *orig;
  int32_t orig_bias, orig_add;
};
// an accepted any-hit test of primitive idx (a sorted slot in the group sweeps) at distance t2
DEVINL void any_accept(Any &a, int idx, float t2) {
  if (a.orig) {
    typedef const int32_t __attribute__((address_space(4))) *ConstI;
    const int id = ((ConstI)(uintptr_t)a.orig)[idx - a.orig_bias] + a.orig_add; // scalar load
    if (a.kocc < 0 || id < a.kocc) {
      a.tocc = t2;
      a.kocc = id;
    }
  } else {
    a.tocc = t2;
    a.kocc = idx;
    a.tb = 0.f;
  }
}
// Each check drains the fetch pipeline (the next block's s_load is re-issued cold), so it is
// taken every 256 primitives, not more often: overshooting an exit by < 256 of 10^4..10^5
// primitives costs far less than a cold scalar load per 32.
constexpr int kExitStride = 32;

template <int PX> DEVINL bool any_looking(const Any (&a)[PX]) {
  bool l = false;
#pragma unroll
  for (int p = 0; p < PX; ++p) l |= a[p].tb > 0.f;
  return __builtin_amdgcn_ballot_w64(l) != 0;
}

template <typename V, int NV>
DEVINL void test_tri_any(const DevTri &T, int idx, const V3<V> (&o)[NV], const V3<V> (&L)[NV],
                         Any (&a)[NV * lanes_of<V>::n]) {
  constexpr int LN = lanes_of<V>::n;
  const f3 e1 = ld3(T.e1), e2 = ld3(T.e2), v0 = ld3(T.v0);
  V det[NV], un[NV], vn[NV];
  V3<V> qv[NV];
  bool any = false;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const V3<V> pv = cross_vu(L[j], e2); // ray_triangle.h:18
    det[j] = dotu(e1, pv);               // :21
    const V3<V> tv = sub_u(o[j], v0);    // :29
    un[j] = dotv(tv, pv);                // :32
    qv[j] = cross_vu(tv, e1);            // :37
    vn[j] = dotv(L[j], qv[j]);           // :40
#pragma unroll
    for (int c = 0; c < LN; ++c)
      any |= tri_candidate(comp(det[j], c), comp(un[j], c), comp(vn[j], c));
  }
  if (ANY_LANE_RARE(any)) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const V tn = dotu(e2, qv[j]); // :45 numerator
#pragma unroll
      for (int c = 0; c < LN; ++c) {
        Any &aa = a[j * LN + c];
        const float de = comp(det[j], c), u = comp(un[j], c), v = comp(vn[j], c);
        float t2, v2;
        if (tri_candidate(de, u, v) && tri_exact(de, u, v, comp(tn, c), aa.tb, t2, v2))
          any_accept(aa, idx, t2);
      }
    }
  }
}

template <typename V, int NV, typename Fetch>
DEVINL void anyhit_tri(Fetch rec, int n, int base, const V3<V> (&o)[NV], const V3<V> (&L)[NV],
                       Any (&a)[NV * lanes_of<V>::n]) {
  for (int k0 = 0; k0 < n; k0 += kExitStride) {
    if (!any_looking(a)) return; // every ray done
    const int m = min(kExitStride, n - k0);
    const int m2 = m & ~1;
    if (m2) {
      DevTri A = rec(k0), B;
      for (int k = 0; k < m2; k += 2) {
        B = rec(rec.landed(A.e2[2], k0 + k + 1));
        __builtin_amdgcn_sched_barrier(0);
        test_tri_any<V, NV>(A, base + k0 + k, o, L, a);
        A = rec(rec.landed(B.e2[2], k0 + min(k + 2, m - 1)));
        __builtin_amdgcn_sched_barrier(0);
        test_tri_any<V, NV>(B, base + k0 + k + 1, o, L, a);
      }
    }
    if (m2 < m) test_tri_any<V, NV>(rec(k0 + m2), base + k0 + m2, o, L, a);
  }
}

// (the ray's filter form is shared by the sphere filter further down, which documents it)
struct RayF { // one shadow ray in filter form, as register pairs for op_sel broadcasts
  v2f o2xy;  // (2ax, 2ay)
  v2f o2z_n; // (2az, nko)
  v2f Lxy;   // (Lx, Ly)
  v2f Lz_s;  // (Lz, ms)
};
DEVINL RayF make_ray_filter(f3 o, f3 L, const float (&g)[3]) {
  const float ax = o.x - g[0], ay = o.y - g[1], az = o.z - g[2];
  const float n = (ax * ax + ay * ay) + az * az;
  const float s = (ax * L.x + ay * L.y) + az * L.z;
  RayF r;
  r.o2xy = v2f{ax + ax, ay + ay};
  r.o2z_n = v2f{az + az, n * -0.9999847412109375f}; // -(1 - 2^-16)
  r.Lxy = v2f{L.x, L.y};
  r.Lz_s = v2f{L.z, -s};
  return r;
}

struct TriPairPF { // DevTriPairPF as eight aligned pairs
  v2f x, y, z, k, gx, gy, gz, pad;
};

// 2 pair records (4 triangles) x 1 ray: q[i] = bounding-sphere test in the form of
// pair4_any_filter_pk (>= 0: the ray may pass within R of the centroid), g[i] = L . n1 / tau'
DEVINL void tripair2_any_prefilter_pk(const TriPairPF (&R)[2], const RayF &r, v2f (&q)[2],
                                      v2f (&g)[2]) {
  v2f x0, x1;
  asm("v_pk_fma_f32 %[y0], %[r0x], %[oxy], %[ozn] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1x], %[oxy], %[ozn] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0x], %[lxy], %[lzs] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1x], %[lxy], %[lzs] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_mul_f32 %[g0], %[r0gx], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[g1], %[r1gx], %[lxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %[y0], %[r0y], %[oxy], %[y0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1y], %[oxy], %[y1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0y], %[lxy], %[x0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1y], %[lxy], %[x1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[g0], %[r0gy], %[lxy], %[g0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[g1], %[r1gy], %[lxy], %[g1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0z], %[ozn], %[y0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1z], %[ozn], %[y1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0z], %[lzs], %[x0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1z], %[lzs], %[x1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[g0], %[r0gz], %[lzs], %[g0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[g1], %[r1gz], %[lzs], %[g1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y0], %[x0], %[x0], %[y0]\n\t"
      "v_pk_fma_f32 %[y1], %[x1], %[x1], %[y1]\n\t"
      "s_nop 1\n\t"
      "v_pk_add_f32 %[y0], %[y0], %[r0k]\n\t"
      "v_pk_add_f32 %[y1], %[y1], %[r1k]\n\t"
      "s_nop 0"
      : [y0] "=&v"(q[0]), [y1] "=&v"(q[1]), [g0] "=&v"(g[0]), [g1] "=&v"(g[1]), [x0] "=&v"(x0),
        [x1] "=&v"(x1)
      : [oxy] "v"(r.o2xy), [ozn] "v"(r.o2z_n), [lxy] "v"(r.Lxy), [lzs] "v"(r.Lz_s),
        [r0x] "s"(R[0].x), [r0y] "s"(R[0].y), [r0z] "s"(R[0].z), [r0k] "s"(R[0].k),
        [r0gx] "s"(R[0].gx), [r0gy] "s"(R[0].gy), [r0gz] "s"(R[0].gz), [r1x] "s"(R[1].x),
        [r1y] "s"(R[1].y), [r1z] "s"(R[1].z), [r1k] "s"(R[1].k), [r1gx] "s"(R[1].gx),
        [r1gy] "s"(R[1].gy), [r1gz] "s"(R[1].gz));
}

// any-hit over triangles [0, n) of `rece` through the filter (1 ray per lane): 2 pair records
// (4 triangles) per step; a step in which any lane has a candidate -- or in which a lane's ray
// starts outside the region the margins were computed for (`far`) -- runs the reference
// arithmetic (test_tri_any) on the exact records, in index order.  `recf` holds ceil(n/2) pair
// records; n_first = index of triangle 0 of this range (even).
constexpr int kTriFilterExit = 128; // triangles between exit checks
template <typename FetchP, typename FetchF, typename FetchE>
DEVINL void anyhit_tri_filter(FetchP recp, FetchF recf, FetchE rece, int n, int base, f3 o, f3 L,
                              const RayF &rs, const RayTF &rf, bool far, Any (&a)[1]) {
  const V3<float> ov[1] = {{o.x, o.y, o.z}}, Lv[1] = {{L.x, L.y, L.z}};
  const int force = far ? 0 : -1; // a far ray is a candidate for every triangle, at both levels
  auto exact = [&](int k, int cnt) {
    for (int i = 0; i < cnt; ++i) test_tri_any<float, 1>(rece(k + i), base + k + i, ov, Lv, a);
  };
  auto level2 = [&](int k) { // triangles k .. k+3 (two pair records): the triangle filter
    const int r = k >> 1;
    const TriPairF R[2] = {recf(r), recf(r + 1)};
    v2f A[2], B[2], C[2];
    tripair2_any_filter_pk(R, rf, A, B, C);
    const int f0 = tri_flags(A[0], B[0], C[0], force), f1 = tri_flags(A[1], B[1], C[1], force);
    if (ANY_LANE_RARE(max(f0, f1) >= 0)) {
      if (__builtin_amdgcn_ballot_w64(f0 >= 0)) exact(k, 2);
      if (__builtin_amdgcn_ballot_w64(f1 >= 0)) exact(k + 2, 2);
    }
  };
  auto test4 = [&](const TriPairPF(&R)[2], int k) { // pre-filter: sphere test or nearly parallel
    v2f q[2], g[2];
    tripair2_any_prefilter_pk(R, rs, q, g);
    const int mq = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                       __float_as_int(q[1].y));
    float mn = 2.f;
    asm("v_min3_f32 %0, %0, |%1|, |%2|\n\tv_min3_f32 %0, %0, |%3|, |%4|"
        : "+v"(mn)
        : "v"(g[0].x), "v"(g[0].y), "v"(g[1].x), "v"(g[1].y));
    if (ANY_LANE_RARE((mq >= 0) | (mn <= 1.f) | far)) level2(k);
  };
  for (int k0 = 0; k0 < n; k0 += kTriFilterExit) {
    if (!__builtin_amdgcn_ballot_w64(a[0].tb > 0.f)) return;
    const int m = min(kTriFilterExit, n - k0);
    const int m4 = m & ~3;
    // one register set, no hand prefetch: the slow paths behind it (64 + 24 more SGPRs) leave no
    // room for a second set (measured: with two, hipcc spilled the prefetched records to VGPR
    // lanes inside this loop); the other waves of the SIMD cover the scalar-load latency
    for (int k = 0; k < m4; k += 4) {
      const int r = (k0 + k) >> 1;
      const TriPairPF R[2] = {recp(r), recp(r + 1)};
      test4(R, k0 + k);
    }
    if (m4 < m) exact(k0 + m4, m - m4);
  }
}

// Triangle GROUPS for shadow rays ("Triangle GROUPS" above; the order of the tests is free for the
// last light, and made free for earlier ones by any_accept's lowest-index rule).  Group records
// have the form of the
// pre-filter's pair records: the bounding sphere in q' form with R = rgeo + 8u at, the cone axis
// over kappa' = (smax + b0 + b1 at + 2^-20) * 1.0001, at = rho_max + |C - g|_1 + rext >= |O - v0_t|_1
// for every member and every ray that starts within rho_max of g (host, rt_capi.cpp commit());
// rays from further out (`far`) open everything.  4 hyper-groups per step; an opened hyper-group
// costs two steps over its 8 super-groups, an opened super-group four over its 16 groups, an
// opened group runs anyhit_tri_filter over its 8 triangles.
// Returns the filter tests this wave swept (4 per step at any level); n_open counts the 8-record
// openings each ray itself needed.
constexpr int kTriGroupExitSteps = 4; // exit check every 16 hyper-groups
template <typename FetchP, typename FetchF, typename FetchE>
DEVINL int anyhit_tri_groups_filter(FetchP recy, FetchP recu, FetchP recg, FetchP recp, FetchF recf,
                                    FetchE rece, int n_hyp, int base, f3 o, f3 L, const RayF &rs,
                                    const RayTF &rf, bool far, Any (&a)[1], int &n_open) {
  int swept = 0;
  auto members = [&](int g) { // sorted triangles [8 g, 8 g + 8) = pair records [4 g, 4 g + 4)
    anyhit_tri_filter(FetchP{recp.p + 4 * g}, FetchF{recf.p + 4 * g}, FetchE{rece.p + 8 * g}, kTriGroup,
                      base + 8 * g, o, L, rs, rf, far, a);
  };
  // 2 pair records = 4 bounding spheres + cones: wave-uniform mask of the ones a LIVE lane may touch.
  // As in the sphere sweep, the group tests see the ray through `rg`: ray-side constant +inf for a
  // `far` ray (every sphere part a candidate), -inf for a lane without a ray; such a lane can still
  // raise the flag through the cone part and is sorted out behind it.
  RayF rg = rs;
  rg.o2z_n.y = !(a[0].tb > 0.f) ? -__builtin_huge_valf() : (far ? __builtin_huge_valf() : rs.o2z_n.y);
  auto open_mask = [&](const TriPairPF(&R)[2]) -> uint32_t {
    v2f q[2], g[2];
    tripair2_any_prefilter_pk(R, rg, q, g);
    const int mq = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                       __float_as_int(q[1].y));
    float mn = 2.f;
    asm("v_min3_f32 %0, %0, |%1|, |%2|\n\tv_min3_f32 %0, %0, |%3|, |%4|"
        : "+v"(mn)
        : "v"(g[0].x), "v"(g[0].y), "v"(g[1].x), "v"(g[1].y));
    uint32_t mask = 0;
    if (ANY_LANE_RARE((mq >= 0) | (mn <= 1.f))) {
      const bool live = a[0].tb > 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool c0 = live && (__float_as_int(q[j].x) >= 0 || fabsf(g[j].x) <= 1.f);
        const bool c1 = live && (__float_as_int(q[j].y) >= 0 || fabsf(g[j].y) <= 1.f);
        if (__builtin_amdgcn_ballot_w64(c0)) mask |= 1u << (2 * j);
        if (__builtin_amdgcn_ballot_w64(c1)) mask |= 2u << (2 * j);
        n_open += (int)c0 + (int)c1;
      }
    }
    return mask;
  };
  auto groups = [&](int s) { // the kTriSuper groups of super-group s, 2 pair records (4 groups) at a time
    for (int q4 = 0; q4 < kTriSuper; q4 += 4) {
      const int r = (kTriSuper * s + q4) >> 1;
      const TriPairPF G[2] = {recg(r), recg(r + 1)};
      uint32_t mask = open_mask(G);
      swept += 4;
      while (mask) {
        const int j = __builtin_ctz(mask);
        mask &= mask - 1;
        members(kTriSuper * s + q4 + j);
        swept += kTriGroup;
      }
    }
  };
  auto supers = [&](int y) { // the kTriHyper super-groups of hyper-group y, 2 pair records at a time
    for (int q4 = 0; q4 < kTriHyper; q4 += 4) {
      const int r = (kTriHyper * y + q4) >> 1;
      const TriPairPF U[2] = {recu(r), recu(r + 1)};
      uint32_t mask = open_mask(U);
      swept += 4;
      while (mask) {
        const int j = __builtin_ctz(mask);
        mask &= mask - 1;
        groups(kTriHyper * y + q4 + j);
      }
    }
  };
  for (int y0 = 0; y0 < n_hyp; y0 += 4 * kTriGroupExitSteps) {
    if (!__builtin_amdgcn_ballot_w64(a[0].tb > 0.f)) return swept;
    const int m = min(4 * kTriGroupExitSteps, n_hyp - y0); // multiple of 4
    swept += m;
    // one register set, as in anyhit_tri_filter: the slow paths behind it leave no room for two
    for (int y = 0; y < m; y += 4) {
      const TriPairPF Y[2] = {recy((y0 + y) >> 1), recy(((y0 + y) >> 1) + 1)};
      uint32_t mask = open_mask(Y);
      while (mask) {
        const int j = __builtin_ctz(mask);
        mask &= mask - 1;
        supers(y0 + y + j);
      }
    }
  }
  return swept;
}

template <typename V, int NV, int NB>
DEVINL void test_sph_any(const DevSph (&s)[NB], int idx, const V3<V> (&o)[NV],
                         const V3<V> (&L)[NV], Any (&a)[NV * lanes_of<V>::n]) {
  constexpr int LN = lanes_of<V>::n;
  V b[NV][NB], q[NV][NB];
  float m = -1.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const V3<V> oc = sub_u(o[j], mk(s[i].cx, s[i].cy, s[i].cz));
      b[j][i] = dotv(oc, L[j]);
      q[j][i] = b[j][i] * b[j][i] - (dotv(oc, oc) - s[i].r2);
#pragma unroll
      for (int c = 0; c < LN; ++c) m = fmaxf(m, comp(q[j][i], c));
    }
  if (ANY_LANE_RARE(!(m < 0.f))) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int c = 0; c < LN; ++c)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          Any &aa = a[j * LN + c];
          float t2;
          if (sph_exact(comp(b[j][i], c), comp(q[j][i], c), aa.tb, t2)) {
            aa.tocc = t2;
            aa.kocc = idx + i;
            aa.tb = 0.f;
          }
        }
  }
}

template <typename V, int NV, typename Fetch>
DEVINL void anyhit_sph(Fetch rec, int n, int base, const V3<V> (&o)[NV], const V3<V> (&L)[NV],
                       Any (&a)[NV * lanes_of<V>::n]) {
  // 4 spheres per fetch (one s_load_dwordx16): scalar loads return out of order, so only ONE
  // fetch can be in flight behind the one being consumed; a longer block hides more latency
  constexpr int NB = (NV * lanes_of<V>::n == 1) ? 4 : 2;
  for (int k0 = 0; k0 < n; k0 += kExitStride) {
    if (!any_looking(a)) return;
    const int m = min(kExitStride, n - k0);
    const int mb = m - m % (2 * NB);
    if (mb) {
      DevSph A[NB], B[NB];
      fetch_batch(rec, k0, A);
      for (int k = 0; k < mb; k += 2 * NB) {
        fetch_batch(rec, rec.landed(A[NB - 1].r2, k0 + k + NB), B);
        test_sph_any<V, NV, NB>(A, base + k0 + k, o, L, a);
        fetch_batch(rec, rec.landed(B[NB - 1].r2, k0 + min(k + 2 * NB, m - NB)), A);
        test_sph_any<V, NV, NB>(B, base + k0 + k + NB, o, L, a);
      }
    }
    for (int k = mb; k < m; ++k) {
      const DevSph s0[1] = {rec(k0 + k)};
      test_sph_any<V, NV, 1>(s0, base + k0 + k, o, L, a);
    }
  }
}

// any-hit over pair records (1 pixel per lane, two spheres per packed op)
constexpr int kPairExitRecords = 128; // exit check every 256 spheres (it drains the fetch pipeline)
template <typename Fetch>
DEVINL int anyhit_sph_pairs(Fetch rec, int n_rec, int base, f3 o, f3 L, Any &a) {
  int swept = 0; // pair records this wave actually tested (wave-uniform)
  const v2f oxy = {o.x, o.y}, oz_ = {o.z, 0.f}, Lxy = {L.x, L.y}, Lz_ = {L.z, 0.f};
  auto accept = [&](const v2f(&b)[2], const v2f(&q)[2], int idx, int nrec) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= nrec) break;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float t2;
        if (sph_exact(comp(b[i], c), comp(q[i], c), a.tb, t2)) {
          a.tocc = t2;
          a.kocc = idx + 2 * i + c;
          a.tb = 0.f;
        }
      }
    }
  };
  // 8 spheres (4 records) per iteration: two packed bodies, ONE candidate filter, one branch.
  // No hand prefetch here: other waves cover the scalar-load latency, and a single register
  // set leaves room for the 32 SGPRs the four records need.
  auto test4 = [&](const PairG(&R0)[2], const PairG(&R1)[2], int idx) {
    v2f b0[2], q0[2], b1[2], q1[2];
    pair2_any_pk(R0, oxy, oz_, Lxy, Lz_, b0, q0);
    pair2_any_pk(R1, oxy, oz_, Lxy, Lz_, b1, q1);
    int m = max(max3i(__float_as_int(q0[0].x), __float_as_int(q0[0].y), __float_as_int(q0[1].x)),
                __float_as_int(q0[1].y));
    m = max3i(m, __float_as_int(q1[0].x), __float_as_int(q1[0].y));
    m = max3i(m, __float_as_int(q1[1].x), __float_as_int(q1[1].y));
    if (ANY_LANE_RARE(m >= 0)) {
      accept(b0, q0, idx, 2);
      accept(b1, q1, idx + 4, 2);
    }
  };
  for (int k0 = 0; k0 < n_rec; k0 += kPairExitRecords) {
    if (!__builtin_amdgcn_ballot_w64(a.tb > 0.f)) return swept;
    const int m = min(kPairExitRecords, n_rec - k0);
    swept += m;
    const int m4 = m & ~3;
    for (int k = 0; k < m4; k += 4) {
      const PairG R0[2] = {rec(k0 + k), rec(k0 + k + 1)};
      const PairG R1[2] = {rec(k0 + k + 2), rec(k0 + k + 3)};
      test4(R0, R1, base + 2 * (k0 + k));
    }
    for (int k = m4; k < m; ++k) {
      const PairG R[2] = {rec(k0 + k), rec(k0 + k)};
      v2f b[2], q[2];
      pair2_any_pk(R, oxy, oz_, Lxy, Lz_, b, q);
      accept(b, q, base + 2 * (k0 + k), 1);
    }
  }
  return swept;
}


// ---------------------------------------------------------------------------------------
// Shadow-ray FILTER (see "FILTERS" above).  Reference, per (ray O,L ; sphere C,r2), 16 operations:
//   oc = fl(O - C); b = fl-dot(oc, L); cc = fl(fl-dot(oc, oc) - r2); disc = fl(fl(b b) - cc).
// With w = O - C (real): |b - w.L| <= 4.01u |w|_1, cc >= |w|^2 - r2 - 6.03u |w|^2 - u r2, so
// "not disc < 0" implies  T(w) := (w.L)^2 - |w|^2 + r2 >= -(21.03u |w|^2 + u r2 + 2^-149).
//
// The filter works relative to a per-scene point g (RenderParams::shadow_center; keeps the
// magnitudes at the scene's size instead of its distance from the world origin):
//   per ray    a = fl(O - g), o2 = 2a, nko = fl(-fl-dot(a,a) (1 - 2^-16)), ms = -fl-dot(a, L)
//   per sphere c = fl(C - g), km = r2 - |c|^2 + 2^-16 (|c|^2 + r2) + 2^-120, rounded up (host, double)
//   y = fma(cz,o2z, fma(cy,o2y, fma(cx,o2x, nko)))        ~ 2 a.c - |a|^2 (+ ray margin)
//   x = fma(cz,Lz,  fma(cy,Ly,  fma(cx,Lx,  ms )))        ~ c.L - a.L = -(w'.L),  w' = a - c
//   q' = fl(fma(x, x, y) + km)                            ~ T(w') + margins        8 operations
// Error budget, with s2 = |a|^2 + |c|^2:  w' differs from w by the roundings of a and c,
// |T(w') - T(w)| <= 13.9u s2 and |w|^2 <= 2.0001 s2, so an accepting pair has
// T(w') >= -(56u s2 + u r2 + 2^-149); the filter's own roundings lose at most 39.4u s2 + 4.01u |a|^2
// (x: 8.3u (|a|+|c|) absolute -> 33.4u s2 on x^2; y: 3.01u (|a|^2 + s2); the fma: u (3 s2 + |a|^2)).
// Needed: 99.4u s2 + 4.01u |a|^2 + u r2 + 2^-149.  Provided: >= 251u |a|^2 by nko (2^-16 = 256u, less
// the 4.02u its own evaluation can lose) and >= 255u (|c|^2 + r2) + 2^-120 by km.  q' is the
// rounding of (fma + km): sign-exact.  Hence  reference accepts  ==>  q' >= 0.
// ---------------------------------------------------------------------------------------
struct PairF { // DevSphPairF as four aligned pairs
  v2f x, y, z, k;
};
// 4 pair records (8 spheres) x 1 ray: q[i] = (x_i^2 + y_i) + km_i; 8 independent chains, so no
// v_pk result is consumed within the next 3 instructions (rule measured in tools/ubench)
DEVINL void pair4_any_filter_pk(const PairF (&R)[4], const RayF &r, v2f (&q)[4]) {
  v2f x0, x1, x2, x3;
  asm("v_pk_fma_f32 %[y0], %[r0x], %[oxy], %[ozn] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1x], %[oxy], %[ozn] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y2], %[r2x], %[oxy], %[ozn] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y3], %[r3x], %[oxy], %[ozn] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0x], %[lxy], %[lzs] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1x], %[lxy], %[lzs] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x2], %[r2x], %[lxy], %[lzs] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x3], %[r3x], %[lxy], %[lzs] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0y], %[oxy], %[y0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1y], %[oxy], %[y1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y2], %[r2y], %[oxy], %[y2] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y3], %[r3y], %[oxy], %[y3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0y], %[lxy], %[x0] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1y], %[lxy], %[x1] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x2], %[r2y], %[lxy], %[x2] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[x3], %[r3y], %[lxy], %[x3] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
      "v_pk_fma_f32 %[y0], %[r0z], %[ozn], %[y0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y1], %[r1z], %[ozn], %[y1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y2], %[r2z], %[ozn], %[y2] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y3], %[r3z], %[ozn], %[y3] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x0], %[r0z], %[lzs], %[x0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x1], %[r1z], %[lzs], %[x1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x2], %[r2z], %[lzs], %[x2] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[x3], %[r3z], %[lzs], %[x3] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[y0], %[x0], %[x0], %[y0]\n\t"
      "v_pk_fma_f32 %[y1], %[x1], %[x1], %[y1]\n\t"
      "v_pk_fma_f32 %[y2], %[x2], %[x2], %[y2]\n\t"
      "v_pk_fma_f32 %[y3], %[x3], %[x3], %[y3]\n\t"
      "v_pk_add_f32 %[y0], %[y0], %[r0k]\n\t"
      "v_pk_add_f32 %[y1], %[y1], %[r1k]\n\t"
      "v_pk_add_f32 %[y2], %[y2], %[r2k]\n\t"
      "v_pk_add_f32 %[y3], %[y3], %[r3k]\n\t"
      "s_nop 0"
      : [y0] "=&v"(q[0]), [y1] "=&v"(q[1]), [y2] "=&v"(q[2]), [y3] "=&v"(q[3]), [x0] "=&v"(x0),
        [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3)
      : [oxy] "v"(r.o2xy), [ozn] "v"(r.o2z_n), [lxy] "v"(r.Lxy), [lzs] "v"(r.Lz_s),
        [r0x] "s"(R[0].x), [r0y] "s"(R[0].y), [r0z] "s"(R[0].z), [r0k] "s"(R[0].k),
        [r1x] "s"(R[1].x), [r1y] "s"(R[1].y), [r1z] "s"(R[1].z), [r1k] "s"(R[1].k),
        [r2x] "s"(R[2].x), [r2y] "s"(R[2].y), [r2z] "s"(R[2].z), [r2k] "s"(R[2].k),
        [r3x] "s"(R[3].x), [r3y] "s"(R[3].y), [r3z] "s"(R[3].z), [r3k] "s"(R[3].k));
}

// any-hit over pair records through the filter: 4 records (8 spheres) per step, double-buffered
// scalar fetches; a step in which any lane has q' >= 0 re-reads its EXACT records and runs the
// reference arithmetic (pair2_any_pk + sph_exact) on them, in index order.  Returns the records
// this wave swept (wave-uniform).
constexpr int kFilterExitRecords = 128; // exit check every 256 spheres
template <typename FetchF, typename FetchE>
DEVINL int anyhit_sph_pairs_filter(FetchF recf, FetchE rece, int n_rec, int base, f3 o, f3 L,
                                   const RayF &rf, Any &a) {
  int swept = 0;
  const v2f oxy = {o.x, o.y}, oz_ = {o.z, 0.f}, Lxy = {L.x, L.y}, Lz_ = {L.z, 0.f};
  auto exact2 = [&](int k, int nrec) { // records k, k+1 (nrec <= 2): the reference arithmetic
    const PairG R[2] = {rece(k), rece(k + nrec - 1)};
    v2f b[2], q[2];
    pair2_any_pk(R, oxy, oz_, Lxy, Lz_, b, q);
    // most filter candidates are margin, not hits: skip the accept code unless a disc is >= 0
    const int m = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                      __float_as_int(q[1].y));
    if (!ANY_LANE_RARE(m >= 0)) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= nrec) break;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float t2;
        if (sph_exact(comp(b[i], c), comp(q[i], c), a.tb, t2)) {
          a.tocc = t2;
          a.kocc = base + 2 * (k + i) + c;
          a.tb = 0.f;
        }
      }
    }
  };
  auto exact4 = [&](int k, int nrec) { // records k .. k+nrec-1 (nrec <= 4), index order
    for (int j = 0; j < nrec; j += 2) exact2(k + j, min(2, nrec - j));
  };
  auto test4 = [&](const PairF(&R)[4], int k) {
    v2f q[4];
    pair4_any_filter_pk(R, rf, q);
    const int m01 = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                        __float_as_int(q[1].y));
    const int m23 = max(max3i(__float_as_int(q[2].x), __float_as_int(q[2].y), __float_as_int(q[3].x)),
                        __float_as_int(q[3].y));
    // a decided / dead lane (tb == 0) may still raise the flag; the exact code accepts nothing for it
    if (ANY_LANE_RARE(max(m01, m23) >= 0)) {
      if (__builtin_amdgcn_ballot_w64(m01 >= 0)) exact2(k, 2);     // index order: records k, k+1
      if (__builtin_amdgcn_ballot_w64(m23 >= 0)) exact2(k + 2, 2); // then k+2, k+3
    }
  };
  for (int k0 = 0; k0 < n_rec; k0 += kFilterExitRecords) {
    if (!__builtin_amdgcn_ballot_w64(a.tb > 0.f)) return swept;
    const int m = min(kFilterExitRecords, n_rec - k0);
    swept += m;
    const int m4 = m & ~3;
    if (m4) {
      PairF A[4], B[4];
      fetch_batch(recf, k0, A);
      for (int k = 0; k < m4; k += 8) {
        fetch_batch(recf, recf.landed(A[3].k, k0 + min(k + 4, m4 - 4)), B);
        test4(A, k0 + k);
        if (k + 4 >= m4) break;
        fetch_batch(recf, recf.landed(B[3].k, k0 + min(k + 8, m4 - 4)), A);
        test4(B, k0 + k + 4);
      }
    }
    if (m4 < m) exact4(k0 + m4, m - m4); // < 4 records left: straight to the reference arithmetic
  }
  return swept;
}

// ---------------------------------------------------------------------------------------
// Sphere GROUPS for shadow rays (rt_device.h SphGroups).  The last light's rays stop at any
// occluder; earlier lights' rays keep the accepted sphere with the lowest original index
// (any_accept above): either way the order of the tests is free.  The bounding sphere (C, R) of a group goes through the
// same q' as a single sphere; R must hold every member's reach.  Member i not rejected at
// `disc < 0` gives T(w_i) >= -(21.03u |w_i|^2 + u r_i^2 + 2^-149) (above; w_i = O - c_i), i.e. the
// line's squared distance from c_i is |w_i|^2 - (w_i.L)^2 / |L|^2 <= r_i^2 (1+u) + 29.1u |w_i|^2 + 2^-149
// (|L|^2 within 8u of 1) and its distance <= r_i (1+u) + 5.4 sqrt(u) |w_i| + 2^-74.  For a ray with
// |fl(O - g)|_1 <= rho_max, |w_i| <= rho_max (1+u) + |C - g| + rgeo, so with rgeo >= r_i + |c_i - C|
// the line passes C within
//     R := rgeo + 0x1.6p-10 (rho_max + |C - g| + rgeo) + 2^-60      (5.4 * 2^-12 = 1.3184e-3 < 0x1.6p-10)
// (host, double).  "Within R of C" is T_G(w) = (w.L)^2 - |w|^2 + R^2 >= (|w|^2 - R^2)(|L|^2 - 1) >=
// -8u (|w|^2 + R^2): inside the hypothesis the single-sphere budget starts from (8 <= 21.03 on |w|^2;
// 8u R^2 instead of u r^2 is covered by km's 255u R^2), so km built from (C, R^2) by the same formula
// makes q'_G >= 0.  Rays that start further than rho_max from g (quirk S3 can do that) treat every
// group as a candidate.
// Three levels, as for primary rays: 8 hyper-groups per step, an opened hyper-group costs one step
// over its 8 super-groups, an opened super-group one over its 8 groups, an opened group one over
// its 8 spheres.
// Returns the filter tests (8 per step, at any level) this wave swept, for the lane-efficiency
// counter; n_open counts the 8-record openings each ray itself needed.
// ---------------------------------------------------------------------------------------
constexpr int kGroupExitSteps = 2; // exit check every 16 hyper-groups
template <typename FetchF, typename FetchE>
DEVINL int anyhit_sph_groups_filter(FetchF recy, FetchF recu, FetchF recg, FetchF recf, FetchE rece,
                                    int n_hyp, int base, f3 o, f3 L, const RayF &rf, bool far, Any &a,
                                    int &n_open) {
  int swept = 0;
  const v2f oxy = {o.x, o.y}, oz_ = {o.z, 0.f}, Lxy = {L.x, L.y}, Lz_ = {L.z, 0.f};
  auto exact2 = [&](int k) { // sorted pair records k, k+1: the reference arithmetic
    const PairG R[2] = {rece(k), rece(k + 1)};
    v2f b[2], q[2];
    pair2_any_pk(R, oxy, oz_, Lxy, Lz_, b, q);
    const int m = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                      __float_as_int(q[1].y));
    if (!ANY_LANE_RARE(m >= 0)) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float t2;
        if (sph_exact(comp(b[i], c), comp(q[i], c), a.tb, t2))
          any_accept(a, base + 2 * (k + i) + c, t2); // a position in the sorted table
      }
  };
  auto members = [&](int g) { // the 8 spheres = 4 pair records of group g
    PairF S[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) S[i] = recf(4 * g + i);
    v2f q[4];
    pair4_any_filter_pk(S, rf, q);
    const int m01 = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                        __float_as_int(q[1].y));
    const int m23 = max(max3i(__float_as_int(q[2].x), __float_as_int(q[2].y), __float_as_int(q[3].x)),
                        __float_as_int(q[3].y));
    // a decided / dead lane (tb == 0) may still raise the flag; the exact code accepts nothing for it
    if (ANY_LANE_RARE(max(m01, m23) >= 0)) {
      if (__builtin_amdgcn_ballot_w64(m01 >= 0)) exact2(4 * g);
      if (__builtin_amdgcn_ballot_w64(m23 >= 0)) exact2(4 * g + 2);
    }
  };
  // 4 pair records = 8 bounding spheres: wave-uniform mask of the ones some LIVE lane may touch.
  // The group tests see the ray through `rg`: its ray-side constant is +inf for a `far` ray (q' =
  // +inf: every group a candidate) and -inf for a lane that carries no ray (q' = -inf: none), so
  // the common path has no per-lane case analysis; a ray decided during the sweep may still raise
  // the flag and is sorted out behind it.
  RayF rg = rf;
  rg.o2z_n.y = !(a.tb > 0.f) ? -__builtin_huge_valf() : (far ? __builtin_huge_valf() : rf.o2z_n.y);
  auto open_mask = [&](const PairF(&G)[4]) -> uint32_t {
    v2f q[4];
    pair4_any_filter_pk(G, rg, q);
    const int mm = max(max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                           __float_as_int(q[1].y)),
                       max(max3i(__float_as_int(q[2].x), __float_as_int(q[2].y), __float_as_int(q[3].x)),
                           __float_as_int(q[3].y)));
    uint32_t mask = 0;
    if (ANY_LANE_RARE(mm >= 0)) {
      const bool live = a.tb > 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool c0 = live && __float_as_int(q[j].x) >= 0;
        const bool c1 = live && __float_as_int(q[j].y) >= 0;
        if (__builtin_amdgcn_ballot_w64(c0)) mask |= 1u << (2 * j);
        if (__builtin_amdgcn_ballot_w64(c1)) mask |= 2u << (2 * j);
        n_open += (int)c0 + (int)c1; // openings THIS ray needs (esc_counters.anyhit_tests)
      }
    }
    return mask;
  };
  auto groups = [&](int s) { // the 8 groups = 4 pair records of super-group s
    PairF G[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) G[i] = recg(4 * s + i);
    uint32_t mask = open_mask(G);
    while (mask) {
      const int j = __builtin_ctz(mask);
      mask &= mask - 1;
      members(8 * s + j);
      swept += kSphGroup;
    }
  };
  auto supers = [&](int y) { // the 8 super-groups = 4 pair records of hyper-group y
    PairF U[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) U[i] = recu(4 * y + i);
    uint32_t mask = open_mask(U);
    while (mask) {
      const int j = __builtin_ctz(mask);
      mask &= mask - 1;
      groups(8 * y + j);
      swept += kSphSuper;
    }
  };
  auto step = [&](const PairF(&Y)[4], int y0) { // 8 hyper-groups
    uint32_t mask = open_mask(Y);
    while (mask) {
      const int j = __builtin_ctz(mask);
      mask &= mask - 1;
      supers(y0 + j);
      swept += kSphHyper;
    }
  };
  for (int y0 = 0; y0 < n_hyp; y0 += 8 * kGroupExitSteps) {
    if (!__builtin_amdgcn_ballot_w64(a.tb > 0.f)) return swept;
    const int m = min(8 * kGroupExitSteps, n_hyp - y0); // hyper-groups in this stretch, multiple of 8
    swept += m;
    PairF A[4], B[4];
    fetch_batch(recy, y0 >> 1, A);
    for (int g = 0; g < m; g += 16) {
      fetch_batch(recy, recy.landed(A[3].k, (y0 + min(g + 8, m - 8)) >> 1), B);
      step(A, y0 + g);
      if (g + 8 >= m) break;
      fetch_batch(recy, recy.landed(B[3].k, (y0 + min(g + 16, m - 8)) >> 1), A);
      step(B, y0 + g + 8);
    }
  }
  return swept;
}

// ---------------------------------------------------------------------------------------
// re-packing of undecided shadow rays inside a workgroup
//
// A wave runs an any-hit loop until its LAST ray is decided, so rays that found their occluder
// early keep occupying lanes: on c4 only 66 % of the executed lane-tests belong to rays the
// reference would still be testing.  The primitive list is therefore cut into segments; between
// segments the workgroup's 256 rays are re-packed through LDS so that the still-undecided ones
// fill whole waves (wave w takes rays [64w, 64w+64) of the packed list) and the other waves sit
// the segment out.  Every ray still meets the primitives in index order and stops at its first
// accepted one, so kocc / tocc -- and the image -- are unchanged.
// ---------------------------------------------------------------------------------------
struct RepackLds {
  float ox[256], oy[256], oz[256]; // shadow-ray origin (main.cpp:757 `hit`)
  float lx[256], ly[256], lz[256]; // unit direction
  float tb[256];                   // bound; 0 = decided or never looking
  float tocc[256];
  int32_t kocc[256];
  int32_t n_open[256]; // sphere groups opened for this ray (rt_device.h SphGroups; counters only)
  uint16_t list[256]; // packed position -> owning thread
  int32_t wave_cnt[4];
  // per-pixel shading state parked here while the any-hit segments run, so it does not hold
  // VGPRs across them (with it in registers hipcc spilled 6 VGPRs to scratch: +133 MB of HBM
  // writes per 4K frame)
  float keep[8][256]; // N.xyz, r, g, b, t, material index
};
constexpr int kSegTris = 256;     // primitives per segment between re-packs
constexpr int kSegSphPairs = 512; // = 1024 spheres
// group sweeps (rt_device.h SphGroups) are short enough that re-packing between segments only
// costs (c4, first segment 256 / 1024 / 4096 records / one segment: shading 0.576 / 0.555 / 0.526 /
// 0.504 ms): one segment up to 2^20 pair records, whole steps of 8 super-groups (256 records)
constexpr int kSegGroupPairs = 1 << 20;

// all 256 threads; returns the number of rays still looking (workgroup-uniform)
DEVINL int repack_rays(RepackLds &R, int tid) {
  __syncthreads(); // tb / kocc writes of the previous segment
  const bool looking = R.tb[tid] > 0.f;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(looking);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (lane == 0) R.wave_cnt[wave] = __popcll(m);
  __syncthreads();
  int off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int c = R.wave_cnt[w];
    off += (w < wave) ? c : 0;
    total += c;
  }
  if (looking) R.list[off + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)tid;
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(total);
}

} // namespace esc
