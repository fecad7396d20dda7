// rt_kernels.hip -- hand-written gfx950 kernels for the per-pixel render loop of
// pg42819/EscTp1RayTracer (reference paths relative to /root/reference):
//
//   k_prepare_primary  per-frame, per-primitive constants for rays leaving the camera origin
//   k_render<STAGE>    one work-item per pixel: camera.h:31-34 get_ray -> main.cpp:176-192
//                      closest hit -> main.cpp:723-789 normal + per-light shadow ray
//                      (main.cpp:314-329) + Phong -> fp32 RGB and/or PPM-quantised bytes
//
// Arithmetic contract: this file MUST be compiled with -ffp-contract=off (hipcc would
// otherwise fuse a*b+c into v_fma_f32 and flip pixels) and with
// correctly rounded fp32 divide/sqrt (hipcc default).  Every expression is evaluated in the
// order the reference evaluates it.  The one liberty taken: vec.h:95-101 starts its dot
// product from `sum = 0`; the leading `0 +` is dropped here.  That can only turn a -0 result
// into +0, and every dot product on this path is either a sum of squares (never -0), or is
// compared against a positive threshold / multiplied by other terms where +-0 behave alike
// (det, u, v, t numerators; d = dot(N,L) tested with `<= 0`; b of the sphere test is squared
// and its -b +- sqrt fallbacks land on the same side of FLT_EPSILON).
//
// Lanes are pixels, so the closest hit is lane-private; primitives are wave-uniform and
// come either through the scalar cache into SGPRs (STAGE_SMEM) or through an LDS chunk the
// workgroup stages (STAGE_LDS).  Inputs must be finite.
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "rt_device.h"

namespace esc {

#define DEVINL __device__ __forceinline__
// wave-uniform, rarely true: keeps the exact tails out of the hot loops' instruction stream
#define ANY_LANE_RARE(cond) __builtin_expect(__builtin_amdgcn_ballot_w64(cond) != 0, 0)

constexpr int STAGE_SMEM = 1;
constexpr int STAGE_LDS = 2;
constexpr int STAGE_BVH = 3;

struct f3 {
  float x, y, z;
};
DEVINL f3 mk(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
DEVINL f3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
DEVINL f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); } // vec.h:111
DEVINL f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); } // vec.h:115
DEVINL f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }    // vec.h:127
DEVINL f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }    // vec.h:119
DEVINL float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }   // vec.h:95
DEVINL f3 cross(f3 a, f3 b) {                                                   // vec.h:103
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DEVINL f3 normalize(f3 v) { return v / sqrtf(dot(v, v)); } // vec.h:135
DEVINL float length(f3 v) { return sqrtf(dot(v, v)); }     // vec.h:139

// ---------------------------------------------------------------------------------------
// lane vectors.  A work-item carries PX pixels; the hot loops see them as NV values of type V,
// where V = float (1 pixel) or v2f (2 pixels in an even/odd VGPR pair).  Arithmetic on v2f
// compiles to v_pk_mul_f32 / v_pk_add_f32: each half is rounded exactly like the scalar
// instruction (no fusion), so results are bit-identical, while the pair issues in ~1.5x the
// time of one scalar op (tools/ubench/valu_rate.hip: 59 -> 77 Tlane-op/s on MI355X).
// ---------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

template <typename V> struct lanes_of { static constexpr int n = 1; };
template <> struct lanes_of<v2f> { static constexpr int n = 2; };
DEVINL float comp(float v, int) { return v; }
DEVINL float comp(v2f v, int c) { return c ? v.y : v.x; }
DEVINL void set_comp(float &v, int, float x) { v = x; }
DEVINL void set_comp(v2f &v, int c, float x) {
  if (c) v.y = x; else v.x = x;
}

template <typename V> struct V3 {
  V x, y, z;
};
template <typename V> DEVINL V3<V> operator-(V3<V> a, V3<V> b) {
  return V3<V>{a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename V> DEVINL V dotv(V3<V> a, V3<V> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// uniform (per-primitive) operands broadcast to every pixel
template <typename V> DEVINL V dotu(f3 a, V3<V> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <typename V> DEVINL V3<V> sub_u(V3<V> a, f3 b) { return V3<V>{a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename V> DEVINL V3<V> cross_vu(V3<V> a, f3 b) { // cross(a, b), vec.h:103 order
  return V3<V>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// gather PX per-pixel f3 into NV lane vectors
template <typename V, int NV>
DEVINL void pack3(const f3 *src, V3<V> (&dst)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int c = 0; c < lanes_of<V>::n; ++c) {
      const f3 v = src[j * lanes_of<V>::n + c];
      set_comp(dst[j].x, c, v.x);
      set_comp(dst[j].y, c, v.y);
      set_comp(dst[j].z, c, v.z);
    }
}

// ---------------------------------------------------------------------------------------
// exact tails (rare paths)
// ---------------------------------------------------------------------------------------

// ray_triangle.h:21-46 given the fp32 numerators: every reject except the `t2 >= *t` bound.
DEVINL bool tri_exact_nb(float detf, float unum, float vnum, float tnum, float &t2o, float &v2o) {
  const double eps = (double)FLT_EPSILON;
  double det = (double)detf;                    // :21
  if (det > -eps && det < eps) return false;    // :23-25
  double inv_det = 1.0 / det;                   // :26 (1.0f widened)
  float u2 = (float)((double)unum * inv_det);   // :32
  if (u2 < FLT_EPSILON || u2 > 1.0f) return false; // :33
  float v2 = (float)((double)vnum * inv_det);   // :40
  if (v2 < FLT_EPSILON || u2 + v2 > 1.0f) return false; // :41
  float t2 = (float)((double)tnum * inv_det);   // :45
  if (t2 < FLT_EPSILON) return false;           // :46
  t2o = t2;
  v2o = v2;
  return true;
}
// ray_triangle.h:21-54.  Returns true on accept.
DEVINL bool tri_exact(float detf, float unum, float vnum, float tnum, float tbound, float &t2o,
                      float &v2o) {
  float t2, v2;
  if (!tri_exact_nb(detf, unum, vnum, tnum, t2, v2)) return false;
  if (t2 >= tbound) return false;               // :49
  t2o = t2;
  v2o = v2;
  return true;
}

// Conservative fp32 pre-reject for the u/v barycentric tests: true means "cannot be
// rejected cheaply, run tri_exact".  With s = sign(det): u2 < eps whenever unum*s <= 0,
// v2 < eps whenever vnum*s <= 0, and u2 + v2 > 1 whenever |unum + vnum| > |det|*(1+1e-5)
// (the fp32 / f64 roundings involved are < 2e-7 relative).  Never rejects an accept.
DEVINL bool tri_candidate(float detf, float unum, float vnum) {
  const uint32_t db = __float_as_uint(detf);
  const uint32_t sg = ((__float_as_uint(unum) ^ db) | (__float_as_uint(vnum) ^ db));
  const float sum = unum + vnum;
  const float m = fabsf(detf) * 1.00001f;
  return ((int32_t)sg >= 0) && !(fabsf(sum) > m);
}

// sphere extension from b and disc; accept iff all three rejects fail.
DEVINL bool sph_exact_nb(float b, float disc, float &t2o) { // without the `t2 >= *t` bound
  if (disc < 0.f) return false;
  float sq = sqrtf(disc);
  float t2 = -b - sq;
  if (t2 < FLT_EPSILON) t2 = -b + sq;
  if (t2 < FLT_EPSILON) return false;
  t2o = t2;
  return true;
}
DEVINL bool sph_exact(float b, float disc, float tbound, float &t2o) {
  float t2;
  if (!sph_exact_nb(b, disc, t2)) return false;
  if (t2 >= tbound) return false;
  t2o = t2;
  return true;
}

// ---------------------------------------------------------------------------------------
// per-frame constants for primary rays
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_prepare_primary(const DevTri *__restrict__ tri, DevTriP *__restrict__ tri_p, int n_tri,
                  const DevSph *__restrict__ sph, DevSphP *__restrict__ sph_p,
                  DevSphPairP *__restrict__ sph2_p, int n_sph, float ox, float oy, float oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const f3 o = mk(ox, oy, oz);
  if (i < n_tri) {
    const DevTri T = tri[i];
    const f3 e1 = ld3(T.e1), e2 = ld3(T.e2);
    const f3 tv = o - ld3(T.v0);   // ray_triangle.h:29
    const f3 qv = cross(tv, e1);   // :37
    DevTriP P;
    P.e2[0] = e2.x; P.e2[1] = e2.y; P.e2[2] = e2.z;
    P.e1[0] = e1.x; P.e1[1] = e1.y; P.e1[2] = e1.z;
    P.tv[0] = tv.x; P.tv[1] = tv.y; P.tv[2] = tv.z;
    P.qv[0] = qv.x; P.qv[1] = qv.y; P.qv[2] = qv.z;
    P.tnum = dot(e2, qv);          // :45 numerator
    P.pad[0] = P.pad[1] = P.pad[2] = 0.f;
    tri_p[i] = P;
  }
  if (i < n_sph) {
    const DevSph S = sph[i];
    const f3 oc = o - mk(S.cx, S.cy, S.cz);
    DevSphP P;
    P.ocx = oc.x;
    P.ocy = oc.y;
    P.ocz = oc.z;
    P.cc = dot(oc, oc) - S.r2;
    sph_p[i] = P;
    DevSphPairP &Q = sph2_p[i >> 1]; // same values, pair-interleaved (each thread owns a half)
    Q.ocx[i & 1] = P.ocx;
    Q.ocy[i & 1] = P.ocy;
    Q.ocz[i & 1] = P.ocz;
    Q.cc[i & 1] = P.cc;
    if (i == n_sph - 1 && (n_sph & 1)) { // pad half: cc = +inf -> disc = -inf, never a candidate
      Q.ocx[1] = Q.ocy[1] = Q.ocz[1] = 0.f;
      Q.cc[1] = __builtin_huge_valf();
    }
  }
}

// same hoisting for the leaf blocks of the acceleration structure (one thread per block slot)
__global__ void __launch_bounds__(256)
k_prepare_bvh(const DevTri *__restrict__ tri, DevTriP *__restrict__ tri_p, int n_tri,
              const DevSph *__restrict__ sph, DevSphP *__restrict__ sph_p, int n_sph, float ox,
              float oy, float oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const f3 o = mk(ox, oy, oz);
  if (i < n_tri) { // a pad slot is all zeros: e1 = 0 keeps det = 0 in this form too
    const DevTri T = tri[i];
    const f3 e1 = ld3(T.e1), e2 = ld3(T.e2);
    const f3 tv = o - ld3(T.v0);
    const f3 qv = cross(tv, e1);
    DevTriP P;
    P.e2[0] = e2.x; P.e2[1] = e2.y; P.e2[2] = e2.z;
    P.e1[0] = e1.x; P.e1[1] = e1.y; P.e1[2] = e1.z;
    P.tv[0] = tv.x; P.tv[1] = tv.y; P.tv[2] = tv.z;
    P.qv[0] = qv.x; P.qv[1] = qv.y; P.qv[2] = qv.z;
    P.tnum = dot(e2, qv);
    P.pad[0] = P.pad[1] = P.pad[2] = 0.f;
    tri_p[i] = P;
  }
  if (i < n_sph) { // a pad slot has r2 = -inf: cc = +inf, disc = -inf
    const DevSph S = sph[i];
    const f3 oc = o - mk(S.cx, S.cy, S.cz);
    DevSphP P;
    P.ocx = oc.x;
    P.ocy = oc.y;
    P.ocz = oc.z;
    P.cc = dot(oc, oc) - S.r2;
    sph_p[i] = P;
  }
}

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
struct Sph2 { // DevSph: (cx, cy), (cz, r2)
  v2f xy, zr;
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

// shadow rays, 2 spheres x 2 pixels:
//   oc = o - c ; b = (ocx*Lx + ocy*Ly) + ocz*Lz ; cc = ((ocx*ocx + ocy*ocy) + ocz*ocz) - r2 ;
//   q = b*b - cc
DEVINL void sph2_any_pk(const Sph2 (&s)[2], v2f ox, v2f oy, v2f oz, v2f Lx, v2f Ly, v2f Lz,
                        v2f (&b)[2], v2f (&q)[2]) {
  v2f ax, ay, az, bx, by, bz, t0, t1; // oc of sphere A / B, temporaries
  asm("v_pk_add_f32 %[ax], %[ox], %[sAxy] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[bx], %[ox], %[sBxy] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ay], %[oy], %[sAxy] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[by], %[oy], %[sBxy] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[az], %[oz], %[sAzr] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[bz], %[oz], %[sBzr] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      // b = dot(oc, L)
      "v_pk_mul_f32 %[bA], %[ax], %[Lx]\n\t"
      "v_pk_mul_f32 %[bB], %[bx], %[Lx]\n\t"
      "v_pk_mul_f32 %[t0], %[ay], %[Ly]\n\t"
      "v_pk_mul_f32 %[t1], %[by], %[Ly]\n\t"
      "v_pk_add_f32 %[bA], %[bA], %[t0]\n\t"
      "v_pk_add_f32 %[bB], %[bB], %[t1]\n\t"
      "v_pk_mul_f32 %[t0], %[az], %[Lz]\n\t"
      "v_pk_mul_f32 %[t1], %[bz], %[Lz]\n\t"
      "v_pk_add_f32 %[bA], %[bA], %[t0]\n\t"
      "v_pk_add_f32 %[bB], %[bB], %[t1]\n\t"
      // dot(oc, oc)
      "v_pk_mul_f32 %[qA], %[ax], %[ax]\n\t"
      "v_pk_mul_f32 %[qB], %[bx], %[bx]\n\t"
      "v_pk_mul_f32 %[t0], %[ay], %[ay]\n\t"
      "v_pk_mul_f32 %[t1], %[by], %[by]\n\t"
      "v_pk_add_f32 %[qA], %[qA], %[t0]\n\t"
      "v_pk_add_f32 %[qB], %[qB], %[t1]\n\t"
      "v_pk_mul_f32 %[t0], %[az], %[az]\n\t"
      "v_pk_mul_f32 %[t1], %[bz], %[bz]\n\t"
      "v_pk_add_f32 %[qA], %[qA], %[t0]\n\t"
      "v_pk_add_f32 %[qB], %[qB], %[t1]\n\t"
      // cc = dot - r2
      "v_pk_add_f32 %[qA], %[qA], %[sAzr] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[qB], %[qB], %[sBzr] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      // q = b*b - cc
      "v_pk_mul_f32 %[t0], %[bA], %[bA]\n\t"
      "v_pk_mul_f32 %[t1], %[bB], %[bB]\n\t"
      "v_pk_add_f32 %[qA], %[t0], %[qA] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[qB], %[t1], %[qB] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 0"
      : [bA] "=&v"(b[0]), [bB] "=&v"(b[1]), [qA] "=&v"(q[0]), [qB] "=&v"(q[1]), [ax] "=&v"(ax),
        [ay] "=&v"(ay), [az] "=&v"(az), [bx] "=&v"(bx), [by] "=&v"(by), [bz] "=&v"(bz),
        [t0] "=&v"(t0), [t1] "=&v"(t1)
      : [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [Lx] "v"(Lx), [Ly] "v"(Ly), [Lz] "v"(Lz),
        [sAxy] "s"(s[0].xy), [sAzr] "s"(s[0].zr), [sBxy] "s"(s[1].xy), [sBzr] "s"(s[1].zr));
}

// ---- 1 pixel per lane, TWO SPHERES per packed op -------------------------------------------
// Same idea with the roles swapped: the lane keeps one ray and the two halves of every v_pk op
// hold spheres 2j and 2j+1, whose constants arrive pair-interleaved (DevSphPairP) and feed the
// ops as plain SGPR pairs; the ray's components are broadcast to both halves through op_sel.
// Measured on c4: primary pass 11.4 -> 10.4 ms.  The 32-op shadow body gains nothing by itself
// (13.1 vs 12.6 ms scalar: its v_pk ops run at ~8 cycles instead of ~4, the register pairs hipcc
// hands to an opaque asm collide in the VGPR banks), but it must be packed too: with a packed
// primary pass and a SCALAR shadow pass sharing the SIMDs the frame took 30.9 ms (rocprofv3:
// fewer VALU instructions, +48 % issue stalls), against 23.5 ms packed/packed and 24.0 ms
// scalar/scalar.
struct PairP { // DevSphPairP as four aligned pairs
  v2f x, y, z, c;
};
struct PairG { // DevSphPair
  v2f x, y, z, r;
};

// primary: b = (ocx*dx + ocy*dy) + ocz*dz ; q = b*b - cc, for records R0 (spheres 0,1), R1 (2,3)
DEVINL void pair2_primary_pk(const PairP (&R)[2], v2f dxy, v2f dz_, v2f (&b)[2], v2f (&q)[2]) {
  v2f t0, t1;
  asm("v_pk_mul_f32 %[b0], %[r0x], %[dxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[b1], %[r1x], %[dxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t0], %[r0y], %[dxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[t1], %[r1y], %[dxy] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_mul_f32 %[t0], %[r0z], %[dz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[t1], %[r1z], %[dz] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %[b0], %[b0], %[t0]\n\t"
      "v_pk_add_f32 %[b1], %[b1], %[t1]\n\t"
      "v_pk_mul_f32 %[q0], %[b0], %[b0]\n\t"
      "v_pk_mul_f32 %[q1], %[b1], %[b1]\n\t"
      "v_pk_add_f32 %[q0], %[q0], %[r0c] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[q1], %[q1], %[r1c] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 0"
      : [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [q0] "=&v"(q[0]), [q1] "=&v"(q[1]), [t0] "=&v"(t0),
        [t1] "=&v"(t1)
      : [dxy] "v"(dxy), [dz] "v"(dz_), [r0x] "s"(R[0].x), [r0y] "s"(R[0].y), [r0z] "s"(R[0].z),
        [r0c] "s"(R[0].c), [r1x] "s"(R[1].x), [r1y] "s"(R[1].y), [r1z] "s"(R[1].z),
        [r1c] "s"(R[1].c));
}

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

// closest hit over pair records [0, n_rec): each record = spheres base+2j, base+2j+1
template <typename Fetch>
DEVINL void closest_sph_primary_pairs(Fetch rec, int n_rec, int base, f3 d, Hit &h) {
  const v2f dxy = {d.x, d.y}, dz_ = {d.z, 0.f};
  auto test = [&](const PairP(&R)[2], int idx) {
    v2f b[2], q[2];
    pair2_primary_pk(R, dxy, dz_, b, q);
    const int m = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                      __float_as_int(q[1].y));
    if (ANY_LANE_RARE(m >= 0)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) { // index order: record i, half c
          float t2;
          if (sph_exact(comp(b[i], c), comp(q[i], c), h.t, t2)) {
            h.t = t2;
            h.idx = idx + 2 * i + c;
          }
        }
    }
  };
  const int n4 = n_rec & ~3;
  if (n4) {
    PairP A[2], B[2];
    fetch_batch(rec, 0, A);
    for (int k = 0; k < n4; k += 4) {
      fetch_batch(rec, rec.landed(A[1].c, k + 2), B);
      test(A, base + 2 * k);
      fetch_batch(rec, rec.landed(B[1].c, min(k + 4, n_rec - 2)), A);
      test(B, base + 2 * k + 4);
    }
  }
  for (int k = n4; k < n_rec; ++k) { // < 4 records left: pair each with itself (idempotent)
    const PairP R[2] = {rec(k), rec(k)};
    v2f b[2], q[2];
    pair2_primary_pk(R, dxy, dz_, b, q);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float t2;
      if (sph_exact(comp(b[0], c), comp(q[0], c), h.t, t2)) {
        h.t = t2;
        h.idx = base + 2 * k + c;
      }
    }
  }
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
};
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
        if (tri_candidate(de, u, v) && tri_exact(de, u, v, comp(tn, c), aa.tb, t2, v2)) {
          aa.tocc = t2;
          aa.kocc = idx;
          aa.tb = 0.f;
        }
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

// SMEM + 2 pixels per lane: hand-scheduled packed body, 2 spheres per batch
template <typename Fetch>
DEVINL void anyhit_sph_pk(Fetch rec, int n, int base, const V3<v2f> &o, const V3<v2f> &L,
                          Any (&a)[2]) {
  auto test2 = [&](const Sph2(&S)[2], int idx) {
    v2f b[2], q[2];
    sph2_any_pk(S, o.x, o.y, o.z, L.x, L.y, L.z, b, q);
    const int m = max(max3i(__float_as_int(q[0].x), __float_as_int(q[0].y), __float_as_int(q[1].x)),
                      __float_as_int(q[1].y));
    if (ANY_LANE_RARE(m >= 0)) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float t2;
          if (sph_exact(comp(b[i], c), comp(q[i], c), a[c].tb, t2)) {
            a[c].tocc = t2;
            a[c].kocc = idx + i;
            a[c].tb = 0.f;
          }
        }
    }
  };
  // n is a multiple of 4 here (the caller peels the remainder)
  for (int k0 = 0; k0 < n; k0 += kExitStride) {
    if (!any_looking(a)) return;
    const int m = min(kExitStride, n - k0);
    Sph2 A[2], B[2];
    fetch_batch(rec, k0, A);
    for (int k = 0; k < m; k += 4) {
      fetch_batch(rec, rec.landed(A[1].zr, k0 + k + 2), B);
      test2(A, base + k0 + k);
      fetch_batch(rec, rec.landed(B[1].zr, k0 + min(k + 4, m - 2)), A);
      test2(B, base + k0 + k + 2);
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
// staging front-ends
// ---------------------------------------------------------------------------------------

// SMEM: the table pointer and index are wave-uniform, so hipcc emits s_load_dwordx4/x8/x16
// and the VALU takes the values straight from SGPRs.
template <typename Rec> struct SmemFetch {
  const Rec *__restrict__ p;
  // Read through the CONSTANT address space: with a wave-uniform address hipcc then always
  // selects s_load, also behind barriers / fences, where its "is this global memory ever
  // written in the kernel?" analysis gives up and would fall back to per-lane global_load.
  // The tables are written before the launch and never by k_render.
  DEVINL Rec operator()(int k) const {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef const u4 __attribute__((address_space(4))) *ConstPtr;
    static_assert(sizeof(Rec) % 16 == 0, "records are whole 16-byte pieces");
    const ConstPtr src = (ConstPtr)(uintptr_t)(p + k);
    Rec r;
    u4 *dst = reinterpret_cast<u4 *>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Rec) / 16); ++i) dst[i] = src[i];
    return r;
  }
  // Scalar loads return out of order, so the only wait hipcc can emit is lgkmcnt(0).  Naming
  // one SGPR of the previous batch in an empty asm makes that wait land HERE, before the next
  // batch's s_load is issued, instead of behind it.
  // (Not `volatile`, no memory clobber: a side-effecting asm makes hipcc give up proving the
  // tables are never written and it falls back from s_load to per-lane global_load.)
  // The index of the next fetch is threaded through the same asm so the s_load cannot be
  // hoisted above it.
  DEVINL int landed(float &x, int next_k) const {
    asm("" : "+s"(x), "+s"(next_k));
    return next_k;
  }
  DEVINL int landed(v2f &x, int next_k) const {
    asm("" : "+s"(x), "+s"(next_k));
    return next_k;
  }
};

// LDS: the workgroup copies a chunk of the table into LDS (16 B per lane per step,
// coalesced), then every lane reads record k at the same address (broadcast ds_read_b128).
template <typename Rec> struct LdsFetch {
  const Rec *p;
  DEVINL Rec operator()(int k) const { return p[k]; }
  DEVINL int landed(float &, int next_k) const { return next_k; } // ds_read is in order
};

template <typename Rec>
DEVINL void lds_stage(Rec *lds, const Rec *__restrict__ src, int n) {
  const uint4 *s = reinterpret_cast<const uint4 *>(src);
  uint4 *d = reinterpret_cast<uint4 *>(lds);
  const int n16 = n * (int)(sizeof(Rec) / 16);
  for (int i = threadIdx.x; i < n16; i += blockDim.x) d[i] = s[i];
}

// splitmix64 finaliser over (seed, pixel, light): counter-based stand-in for the
// reference's mt19937 draw at main.cpp:743-747 (the test checker restates the same hash).
DEVINL uint32_t face_hash(uint64_t seed, uint32_t pixel, uint32_t light, uint32_t n_faces) {
  uint64_t z = seed + (((uint64_t)pixel << 32) | (uint64_t)light) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)((z >> 32) % (uint64_t)n_faces);
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
  uint16_t list[256]; // packed position -> owning thread
  int32_t wave_cnt[4];
};
constexpr int kSegTris = 256;     // primitives per segment between re-packs
constexpr int kSegSphPairs = 512; // = 1024 spheres

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

// ---------------------------------------------------------------------------------------
// ESC_STAGE_BVH: wave-synchronous walk of the bounding-volume tree (rt_device.h BvhNode).
//
// The 64 rays of a wave (a 16 x 4 pixel block, or the shadow rays leaving it towards one light)
// travel together: the node index is wave-uniform, the node comes through the scalar cache into
// SGPRs as one s_load_dwordx16, each lane tests its own ray against the two child boxes, and a
// child is entered when ANY lane needs it.  The stack is wave-uniform too and lives in the 64
// lanes of one VGPR (a select to push, v_readlane to pop), so there are no per-lane gathers, no LDS and no
// divergence inside the walk; the price -- a lane rides along through subtrees only its
// neighbours need -- is small for rays this coherent.  Leaves are blocks of primitives tested by
// all lanes with the SAME exact tests as the brute-force loops, so a ray can only ever see a
// subset of the primitives brute force shows it; the box pads (accel_build.cpp) make sure the
// primitives it would accept are never culled.
//
// The box test is NOT part of the reference arithmetic (it only decides what gets tested), so it
// may use fused multiply-adds: t = plane * (1/d) - o * (1/d).
//
// Three walks share the code:
//   MODE 0  closest hit (main.cpp:176-192): bound shrinks, near child first, ties go to the
//           smaller key (= the primitive brute force meets first, ray_triangle.h:49 is strict)
//   MODE 1  any hit (main.cpp:314-329) when the occluder's t2 is not needed afterwards
//   MODE 2  first hit in primitive order: occlusion() returns its FIRST occluder's t2 through the
//           caller's t (quirk S3), and the next light's shadow ray starts from it; subtrees whose
//           smallest key cannot beat the current one are skipped (BvhNode::minkey)
// ---------------------------------------------------------------------------------------
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

struct RaySt {
  float tmax;   // MODE 0: closest t so far (FLT_MAX none); MODE 1/2: the ray's fixed bound
  float thit;   // MODE 1/2: t2 of the accepted occluder
  float v;      // MODE 0: barycentric v of the closest triangle (quirk S1)
  uint32_t key; // accepted primitive, kNoKey = none
};

template <int MODE> DEVINL void offer(RaySt &s, float t2, float v2, uint32_t key) {
  if (MODE == 0) {
    if (t2 < s.tmax || (t2 == s.tmax && s.key != kNoKey && key < s.key)) {
      s.tmax = t2;
      s.v = v2;
      s.key = key;
    }
  } else if (MODE == 1) {
    if (s.key == kNoKey && t2 < s.tmax) {
      s.key = key;
      s.thit = t2;
    }
  } else {
    if (t2 < s.tmax && key < s.key) {
      s.key = key;
      s.thit = t2;
    }
  }
}

struct RayBox {
  f3 inv, noinv; // 1/d and -(o/d), d nudged off zero so both stay finite
};
DEVINL float safe_rcp(float d) {
  const float a = (fabsf(d) < 1e-30f) ? copysignf(1e-30f, d) : d;
  return 1.0f / a;
}
DEVINL RayBox ray_box(f3 o, f3 d) {
  RayBox r;
  r.inv = mk(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
  r.noinv = mk(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
  return r;
}
// ray segment [0, tmax] against a wave-uniform box; tn = entry distance
DEVINL bool slab(const float (&lo)[3], const float (&hi)[3], const RayBox &rb, float tmax,
                 float &tn) {
  const float x0 = __builtin_fmaf(lo[0], rb.inv.x, rb.noinv.x);
  const float x1 = __builtin_fmaf(hi[0], rb.inv.x, rb.noinv.x);
  const float y0 = __builtin_fmaf(lo[1], rb.inv.y, rb.noinv.y);
  const float y1 = __builtin_fmaf(hi[1], rb.inv.y, rb.noinv.y);
  const float z0 = __builtin_fmaf(lo[2], rb.inv.z, rb.noinv.z);
  const float z1 = __builtin_fmaf(hi[2], rb.inv.z, rb.noinv.z);
  tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.f));
  const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
  return tn <= tf;
}

// N wave-uniform triangles against one ray per lane: ray_triangle.h:14-46.  key(i) names slot i
// (only evaluated for an accept).  General form: any origin.
template <int MODE, int N, typename KeyFn>
DEVINL void test_tris_general(const DevTri (&T)[N], KeyFn key, f3 o, f3 d, RaySt &s, bool act) {
  float det[N], un[N], vn[N];
  f3 qv[N];
  bool cand = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f3 e1 = ld3(T[i].e1), e2 = ld3(T[i].e2), v0 = ld3(T[i].v0);
    const f3 pv = cross(d, e2); // :18
    det[i] = dot(e1, pv);       // :21
    const f3 tv = o - v0;       // :29
    un[i] = dot(tv, pv);        // :32
    qv[i] = cross(tv, e1);      // :37
    vn[i] = dot(d, qv[i]);      // :40
    cand |= act && tri_candidate(det[i], un[i], vn[i]);
  }
  if (ANY_LANE_RARE(cand)) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float t2, v2;
      if (act && tri_candidate(det[i], un[i], vn[i]) &&
          tri_exact_nb(det[i], un[i], vn[i], dot(ld3(T[i].e2), qv[i]), t2, v2))
        offer<MODE>(s, t2, v2, key(i));
    }
  }
}
// Primary form: tvec, qvec and dot(edge2,qvec) hoisted per triangle (k_prepare_*), same bits.
template <int MODE, int N, typename KeyFn>
DEVINL void test_tris_primary(const DevTriP (&T)[N], KeyFn key, f3 d, RaySt &s, bool act) {
  float det[N], un[N], vn[N];
  bool cand = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f3 pv = cross(d, ld3(T[i].e2)); // :18
    det[i] = dot(ld3(T[i].e1), pv);       // :21
    un[i] = dot(ld3(T[i].tv), pv);        // :32
    const f3 qv = ld3(T[i].qv);
    vn[i] = (qv.x * d.x + qv.y * d.y) + qv.z * d.z; // :40 (products commute, sum order kept)
    cand |= act && tri_candidate(det[i], un[i], vn[i]);
  }
  if (ANY_LANE_RARE(cand)) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float t2, v2;
      if (act && tri_candidate(det[i], un[i], vn[i]) &&
          tri_exact_nb(det[i], un[i], vn[i], T[i].tnum, t2, v2))
        offer<MODE>(s, t2, v2, key(i));
    }
  }
}

// N wave-uniform spheres (SURVEY.md 8(d) test), general and primary (oc, cc hoisted) forms
template <int MODE, int N, typename KeyFn>
DEVINL void test_sphs_general(const DevSph (&S)[N], KeyFn key, f3 o, f3 d, RaySt &s, bool act) {
  float b[N], q[N];
  float m = -1.f;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f3 oc = o - mk(S[i].cx, S[i].cy, S[i].cz);
    b[i] = dot(oc, d);
    q[i] = b[i] * b[i] - (dot(oc, oc) - S[i].r2);
    m = fmaxf(m, q[i]);
  }
  if (ANY_LANE_RARE(act && !(m < 0.f))) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float t2;
      if (act && sph_exact_nb(b[i], q[i], t2)) offer<MODE>(s, t2, 0.f, key(i));
    }
  }
}
template <int MODE, int N, typename KeyFn>
DEVINL void test_sphs_primary(const DevSphP (&S)[N], KeyFn key, f3 d, RaySt &s, bool act) {
  float b[N], q[N];
  float m = -1.f;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    b[i] = (S[i].ocx * d.x + S[i].ocy * d.y) + S[i].ocz * d.z;
    q[i] = b[i] * b[i] - S[i].cc;
    m = fmaxf(m, q[i]);
  }
  if (ANY_LANE_RARE(act && !(m < 0.f))) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float t2;
      if (act && sph_exact_nb(b[i], q[i], t2)) offer<MODE>(s, t2, 0.f, key(i));
    }
  }
}

// Walks one tree.  `alive`: this lane carries a ray.  Leaf(blk, act) tests a leaf block.
// n_visits counts the nodes + leaves the WAVE went through (wave-uniform).
// Lane predicates are kept as 64-bit wave masks (SGPR pairs) and combined on the scalar unit:
// the box tests run for every lane unconditionally and their ballots are masked afterwards.
template <int MODE, typename Leaf>
DEVINL void bvh_walk(const BvhRef &R, f3 o, f3 d, RaySt &s, bool alive, Leaf leaf, int &n_visits) {
  typedef unsigned long long mask_t;
  const RayBox rb = ray_box(o, d);
  const SmemFetch<BvhNode> nodes{R.nodes};
  const mask_t alive_m = __builtin_amdgcn_ballot_w64(alive);
  int stack = 0; // lane i holds stack entry i
  const int lane_id = (int)(threadIdx.x & 63u);
  int sp = 0;
  int cur = R.root;
  for (;;) {
    while (cur >= 0) {
      ++n_visits;
      const BvhNode N = nodes(cur);
      mask_t act_m = alive_m;
      if (MODE == 1) act_m = __builtin_amdgcn_ballot_w64(alive && s.key == kNoKey);
      float tn0, tn1;
      const bool b0 = slab(N.lo0, N.hi0, rb, s.tmax, tn0);
      const bool b1 = slab(N.lo1, N.hi1, rb, s.tmax, tn1);
      mask_t m0 = __builtin_amdgcn_ballot_w64(b0) & act_m;
      mask_t m1 = __builtin_amdgcn_ballot_w64(b1) & act_m;
      if (MODE == 2) {
        m0 &= __builtin_amdgcn_ballot_w64(N.minkey[0] < s.key);
        m1 &= __builtin_amdgcn_ballot_w64(N.minkey[1] < s.key);
      }
      if (m0 != 0 && m1 != 0) {
        bool one_first = false;
        if (MODE == 0) { // near child first: majority vote of the lanes that care
          const mask_t lt = __builtin_amdgcn_ballot_w64(tn1 < tn0);
          const mask_t both = m0 & m1;
          const mask_t p1 = (m1 & ~m0) | (both & lt);
          const mask_t p0 = (m0 & ~m1) | (both & ~lt);
          one_first = __popcll(p1) > __popcll(p0);
        }
        const int c_far = one_first ? N.child[0] : N.child[1];
        stack = (lane_id == sp) ? c_far : stack; // "v_writelane": one compare + select
        ++sp;
        cur = one_first ? N.child[1] : N.child[0];
      } else if (m0 != 0) {
        cur = N.child[0];
      } else if (m1 != 0) {
        cur = N.child[1];
      } else {
        if (sp == 0) return;
        --sp;
        cur = __builtin_amdgcn_readlane(stack, sp);
      }
    }
    ++n_visits;
    leaf(~cur, (MODE == 1) ? (alive && s.key == kNoKey) : alive);
    if (MODE == 1 && __builtin_amdgcn_ballot_w64(alive && s.key == kNoKey) == 0) return;
    if (sp == 0) return;
    --sp;
    cur = __builtin_amdgcn_readlane(stack, sp);
  }
}

// both trees, triangles first (their keys are smaller: main.cpp:179-186 meets them first)
// n_tests: leaf primitives tested while this LANE was still undecided; n_swept: leaf primitives
// the WAVE tested (wave-uniform).  PRIMARY: every ray starts at the camera, so the leaves are read
// in their hoisted per-frame form (k_prepare_bvh).  A handful of triangles (a floor, a light)
// is not worth a tree: up to kTinyTris are simply tested in index order from the flat tables.
constexpr int kTinyTris = 4;
template <int MODE, bool PRIMARY>
DEVINL void bvh_trace(const RenderParams &p, f3 o, f3 d, RaySt &s, bool alive, int &n_visits,
                      int &n_tests, int &n_swept) {
  if (__builtin_amdgcn_ballot_w64(alive) == 0) return;
  if (p.n_tri > 0 && p.n_tri <= kTinyTris) {
    for (int k = 0; k < p.n_tri; ++k) {
      const bool act = (MODE == 1) ? (alive && s.key == kNoKey) : alive;
      auto key = [&](int) { return (uint32_t)k; };
      if (PRIMARY) {
        const DevTriP T[1] = {SmemFetch<DevTriP>{p.tri_p}(k)};
        test_tris_primary<MODE, 1>(T, key, d, s, act);
      } else {
        const DevTri T[1] = {SmemFetch<DevTri>{p.tri}(k)};
        test_tris_general<MODE, 1>(T, key, o, d, s, act);
      }
      n_tests += act ? 1 : 0;
      n_swept += 1;
    }
  } else if (p.n_tri > 0) {
    const int32_t *order = p.bvh_tri.order;
    bvh_walk<MODE>(p.bvh_tri, o, d, s, alive,
                   [&](int blk, bool act) {
                     auto key = [&](int i) { return (uint32_t)order[blk * kTriBlock + i]; };
                     if (PRIMARY) {
                       const TriBlockP B = SmemFetch<TriBlockP>{
                           reinterpret_cast<const TriBlockP *>(p.bvh_tri.blocks_p)}(blk);
                       test_tris_primary<MODE, kTriBlock>(B.t, key, d, s, act);
                     } else {
                       const TriBlock B = SmemFetch<TriBlock>{
                           reinterpret_cast<const TriBlock *>(p.bvh_tri.blocks)}(blk);
                       test_tris_general<MODE, kTriBlock>(B.t, key, o, d, s, act);
                     }
                     n_tests += act ? kTriBlock : 0;
                     n_swept += kTriBlock;
                   },
                   n_visits);
  }
  if (p.n_sph > 0) {
    if (MODE == 1 && __builtin_amdgcn_ballot_w64(alive && s.key == kNoKey) == 0) return;
    const int32_t *order = p.bvh_sph.order;
    const uint32_t key_base = (uint32_t)p.n_tri;
    bvh_walk<MODE>(p.bvh_sph, o, d, s, alive,
                   [&](int blk, bool act) {
                     auto key = [&](int i) {
                       return key_base + (uint32_t)order[blk * kSphBlock + i];
                     };
                     if (PRIMARY) {
                       const SphBlockP B = SmemFetch<SphBlockP>{
                           reinterpret_cast<const SphBlockP *>(p.bvh_sph.blocks_p)}(blk);
                       test_sphs_primary<MODE, kSphBlock>(B.s, key, d, s, act);
                     } else {
                       const SphBlock B = SmemFetch<SphBlock>{
                           reinterpret_cast<const SphBlock *>(p.bvh_sph.blocks)}(blk);
                       test_sphs_general<MODE, kSphBlock>(B.s, key, o, d, s, act);
                     }
                     n_tests += act ? kSphBlock : 0;
                     n_swept += kSphBlock;
                   },
                   n_visits);
  }
}

// ---------------------------------------------------------------------------------------
// screen-space bins for primary rays (rt_device.h BinGrid)
// ---------------------------------------------------------------------------------------

// One WAVE per primitive: lanes 0..7 project the eight corners of its padded box, shuffles
// reduce them to a pixel rectangle, then the 64 lanes append the primitive to the bins of that
// rectangle side by side (an append is an atomic whose result is needed, so one thread doing
// them in turn is latency bound).  Projection in double: a world point X lies on the primary ray
// of image-plane coordinates (s,t) iff X - o = l * (A + s*hor + t*ver), A = llc - o, l > 0
// (camera.h:31-34), so (l*s, l*t, l) = M^-1 (X - o) with M = [hor ver A].  The rays that meet a
// convex box lying wholly in front of the camera plane are exactly those through the convex hull
// of its projected corners, which the pixel bounding box (grown by one pixel for the fp32
// rounding of main.cpp:709-713) contains.
constexpr int kBinMaxSpan = 2048; // bins one primitive may be appended to before it goes global
__global__ void __launch_bounds__(256)
k_bin_primary(const RenderParams p, const PrimBoxDev *__restrict__ tri_boxes,
              const PrimBoxDev *__restrict__ sph_boxes) {
  const int i = blockIdx.x * 4 + (int)(threadIdx.x >> 6); // primitive of this wave
  const int lane = (int)(threadIdx.x & 63u);
  if (i >= p.n_tri + p.n_sph) return;
  const bool is_sph = i >= p.n_tri;
  const int id = is_sph ? i - p.n_tri : i;
  const PrimBoxDev B = is_sph ? sph_boxes[id] : tri_boxes[id];
  const BinGrid g = p.bins;

  const double o[3] = {p.origin[0], p.origin[1], p.origin[2]};
  const double a[3] = {p.horizontal[0], p.horizontal[1], p.horizontal[2]};
  const double b[3] = {p.vertical[0], p.vertical[1], p.vertical[2]};
  const double c[3] = {(double)p.llc[0] - o[0], (double)p.llc[1] - o[1], (double)p.llc[2] - o[2]};
  const double bxc[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
  const double cxa[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
  const double axb[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  const double det = a[0] * bxc[0] + a[1] * bxc[1] + a[2] * bxc[2];
  const double inv_det = 1.0 / det;

  const int k = lane & 7; // corner (lanes >= 8 repeat them: harmless for min/max/any)
  const double q[3] = {(double)((k & 1) ? B.hi[0] : B.lo[0]) - o[0],
                       (double)((k & 2) ? B.hi[1] : B.lo[1]) - o[1],
                       (double)((k & 4) ? B.hi[2] : B.lo[2]) - o[2]};
  const double ls = (bxc[0] * q[0] + bxc[1] * q[1] + bxc[2] * q[2]) * inv_det;
  const double lt = (cxa[0] * q[0] + cxa[1] * q[1] + cxa[2] * q[2]) * inv_det;
  const double l = (axb[0] * q[0] + axb[1] * q[1] + axb[2] * q[2]) * inv_det;
  const double qm = fmax(fabs(q[0]), fmax(fabs(q[1]), fabs(q[2])));
  const bool front = l > 1e-6 * (1.0 + qm);
  const unsigned long long fm = __builtin_amdgcn_ballot_w64(front) & 0xFFull;
  if (fm == 0) return; // wholly behind the camera plane: no primary ray can reach it
  bool global = fm != 0xFFull; // straddles the camera plane
  int tx0 = 0, tx1 = -1, gy0 = 0, gy1 = -1;
  if (!global) {
    const double wp = ls / l * (double)(p.W - 1), hp = lt / l * (double)(p.H - 1);
    double wmin = wp, wmax = wp, hmin = hp, hmax = hp;
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { // butterfly over the 8 corners
      wmin = fmin(wmin, __shfl_xor(wmin, off));
      wmax = fmax(wmax, __shfl_xor(wmax, off));
      hmin = fmin(hmin, __shfl_xor(hmin, off));
      hmax = fmax(hmax, __shfl_xor(hmax, off));
    }
    const double big = 1e9;
    tx0 = (int)floor(fmax(-big, fmin(big, (wmin - 1.0) / 32.0)));
    tx1 = (int)floor(fmax(-big, fmin(big, (wmax + 1.0) / 32.0)));
    gy0 = (int)floor(fmax(-big, fmin(big, (hmin - 1.0) / (double)kTileH)));
    gy1 = (int)floor(fmax(-big, fmin(big, (hmax + 1.0) / (double)kTileH)));
    tx0 = max(tx0, 0);
    gy0 = max(gy0, 0);
    tx1 = min(tx1, g.tiles_x - 1);
    gy1 = min(gy1, g.groups_y - 1);
    if (tx0 > tx1 || gy0 > gy1) return; // off screen
    global = (long long)(tx1 - tx0 + 1) * (gy1 - gy0 + 1) > kBinMaxSpan;
  }
  if (global) {
    if (lane == 0) {
      const int slot = atomicAdd(&g.hdr[is_sph ? 1 : 0], 1);
      if (slot < kBinGlobalCap) g.hdr[(is_sph ? 2 + kBinGlobalCap : 2) + slot] = id;
    }
    return;
  }
  int32_t *counts = g.hdr + kBinHdrInts;
  int32_t *ids = is_sph ? g.sph_ids : g.tri_ids;
  const int nx = tx1 - tx0 + 1, n = nx * (gy1 - gy0 + 1);
  for (int j = lane; j < n; j += 64) {
    const int bin = (gy0 + j / nx) * g.tiles_x + tx0 + j % nx;
    const int slot = atomicAdd(&counts[2 * bin + (is_sph ? 1 : 0)], 1);
    if (slot < kBinCap) ids[(size_t)bin * kBinCap + slot] = id;
  }
}

// Closest hit of a tile's primary rays from its bin.  Returns false (nothing tested) when the bin
// cannot be used; the caller then walks the tree.  Slots past a bin's count hold ids of earlier
// frames or zeros -- always valid primitives of the current scene, and testing an extra
// primitive cannot change a closest hit -- so lists are read in whole batches.
DEVINL bool bin_trace(const RenderParams &p, int tx, int h_tile, f3 d, RaySt &s, bool alive) {
  const BinGrid g = p.bins;
  if (g.hdr == nullptr || (h_tile % kTileH) != 0) return false;
  const int gy = h_tile / kTileH;
  if (tx >= g.tiles_x || gy >= g.groups_y) return false;
  typedef const int32_t __attribute__((address_space(4))) *CI;
  const CI hdr = (CI)(uintptr_t)g.hdr;
  const int bin = gy * g.tiles_x + tx;
  const int n_gt = hdr[0], n_gs = hdr[1];
  const int n_t = hdr[kBinHdrInts + 2 * bin], n_s = hdr[kBinHdrInts + 2 * bin + 1];
  if (n_gt > kBinGlobalCap || n_gs > kBinGlobalCap || n_t > kBinCap || n_s > kBinCap) return false;
  const SmemFetch<DevTriP> tris{p.tri_p};
  const SmemFetch<DevSphP> sphs{p.sph_p};
  const uint32_t nt = (uint32_t)p.n_tri;
  auto tri_list = [&](CI ids, int n) {
    for (int k = 0; k < n; ++k) {
      const int id = ids[k];
      const DevTriP T[1] = {tris(id)};
      test_tris_primary<0, 1>(T, [&](int) { return (uint32_t)id; }, d, s, alive);
    }
  };
  auto sph_list = [&](CI ids, int n) { // n rounded up to whole batches of 4 by the caller
    for (int k = 0; k < n; k += 4) {
      const int i0 = ids[k], i1 = ids[k + 1], i2 = ids[k + 2], i3 = ids[k + 3];
      const DevSphP S[4] = {sphs(i0), sphs(i1), sphs(i2), sphs(i3)};
      test_sphs_primary<0, 4>(
          S, [&](int i) { return nt + (uint32_t)(i == 0 ? i0 : i == 1 ? i1 : i == 2 ? i2 : i3); }, d,
          s, alive);
    }
  };
  tri_list(hdr + 2, n_gt);
  tri_list((CI)(uintptr_t)(g.tri_ids + (size_t)bin * kBinCap), n_t);
  if (p.n_sph > 0) {
    sph_list(hdr + 2 + kBinGlobalCap, (n_gs + 3) & ~3);
    sph_list((CI)(uintptr_t)(g.sph_ids + (size_t)bin * kBinCap), (n_s + 3) & ~3);
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// light-space bins for shadow rays (rt_device.h LightBins)
// ---------------------------------------------------------------------------------------

// cube-map face of a direction v: 2*axis + (negative ? 1 : 0), axis = the largest |component|
// (lowest index on ties); (u, w) = the other two components over |v[axis]|, in axis order
DEVINL int cube_face(f3 v, float &u, float &w) {
  const float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z);
  int m = 0;
  float dm = ax;
  if (ay > dm) { m = 1; dm = ay; }
  if (az > dm) { m = 2; dm = az; }
  const float vm = (m == 0) ? v.x : (m == 1) ? v.y : v.z;
  const float va = (m == 0) ? v.y : v.x;
  const float vb = (m == 2) ? v.y : v.z;
  u = va / dm;
  w = vb / dm;
  return 2 * m + ((vm < 0.f) ? 1 : 0);
}

// One wave per primitive, once per scene: for every light point and cube face, lanes 0..7 project
// the eight corners of the padded box (double), the wave reduces them to a cell rectangle (grown
// by 1e-5 in face coordinates for the fp32 lookup in k_shade) and appends side by side.
__global__ void __launch_bounds__(256)
k_bin_light(const LightBins g, const float *__restrict__ light_points,
            const PrimBoxDev *__restrict__ tri_boxes, int n_tri,
            const PrimBoxDev *__restrict__ sph_boxes, int n_sph) {
  const int i = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  if (i >= n_tri + n_sph) return;
  const bool is_sph = i >= n_tri;
  const int id = is_sph ? i - n_tri : i;
  const PrimBoxDev B = is_sph ? sph_boxes[id] : tri_boxes[id];
  const int k = lane & 7;
  const int R = g.R;
  for (int pt = 0; pt < g.n_points; ++pt) {
    const double L[3] = {light_points[4 * pt], light_points[4 * pt + 1], light_points[4 * pt + 2]};
    const double q[3] = {(double)((k & 1) ? B.hi[0] : B.lo[0]) - L[0],
                         (double)((k & 2) ? B.hi[1] : B.lo[1]) - L[1],
                         (double)((k & 4) ? B.hi[2] : B.lo[2]) - L[2]};
    const double qm = fmax(fabs(q[0]), fmax(fabs(q[1]), fabs(q[2])));
    for (int face = 0; face < 6; ++face) {
      const int m = face >> 1;
      const double sg = (face & 1) ? -1.0 : 1.0;
      const double depth = sg * q[m];
      const double qa = (m == 0) ? q[1] : q[0], qb = (m == 2) ? q[1] : q[2];
      const bool front = depth > 1e-9 * (1.0 + qm);
      const unsigned long long fm = __builtin_amdgcn_ballot_w64(front) & 0xFFull;
      if (fm == 0) continue; // wholly behind this face's plane through L
      int32_t *hdr = g.face_hdr + (size_t)(pt * 6 + face) * kBinHdrInts;
      bool global = fm != 0xFFull;
      int cu0 = 0, cu1 = -1, cw0 = 0, cw1 = -1;
      if (!global) {
        const double u = qa / depth, w = qb / depth;
        double umin = u, umax = u, wmin = w, wmax = w;
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
          umin = fmin(umin, __shfl_xor(umin, off));
          umax = fmax(umax, __shfl_xor(umax, off));
          wmin = fmin(wmin, __shfl_xor(wmin, off));
          wmax = fmax(wmax, __shfl_xor(wmax, off));
        }
        if (umin > 1.0 + 1e-5 || umax < -1.0 - 1e-5 || wmin > 1.0 + 1e-5 || wmax < -1.0 - 1e-5)
          continue; // seen from L through other faces only
        const double h = 0.5 * (double)R;
        cu0 = max(0, (int)floor((fmax(umin, -1.0) - 1e-5 + 1.0) * h));
        cu1 = min(R - 1, (int)floor((fmin(umax, 1.0) + 1e-5 + 1.0) * h));
        cw0 = max(0, (int)floor((fmax(wmin, -1.0) - 1e-5 + 1.0) * h));
        cw1 = min(R - 1, (int)floor((fmin(wmax, 1.0) + 1e-5 + 1.0) * h));
        global = (long long)(cu1 - cu0 + 1) * (cw1 - cw0 + 1) > kBinMaxSpan;
      }
      if (global) {
        if (lane == 0) {
          const int slot = atomicAdd(&hdr[is_sph ? 1 : 0], 1);
          if (slot < kBinGlobalCap) hdr[(is_sph ? 2 + kBinGlobalCap : 2) + slot] = id;
        }
        continue;
      }
      int32_t *ids = is_sph ? g.sph_ids : g.tri_ids;
      const size_t cell0 = (size_t)(pt * 6 + face) * R * R;
      const int nx = cu1 - cu0 + 1, n = nx * (cw1 - cw0 + 1);
      for (int j = lane; j < n; j += 64) {
        const size_t cell = cell0 + (size_t)(cw0 + j / nx) * R + cu0 + j % nx;
        const int slot = atomicAdd(&g.counts[2 * cell + (is_sph ? 1 : 0)], 1);
        if (slot < kBinCap) ids[cell * kBinCap + slot] = id;
      }
    }
  }
}

// cell of the shadow ray that ends in light point `pt` and starts at `ro`, or -1
DEVINL int light_cell(const LightBins &g, int pt, f3 Lp, f3 ro) {
  float u, w;
  const int face = cube_face(ro - Lp, u, w);
  const float h = 0.5f * (float)g.R;
  const int cu = min(g.R - 1, max(0, (int)floorf((u + 1.f) * h)));
  const int cw = min(g.R - 1, max(0, (int)floorf((w + 1.f) * h)));
  return ((pt * 6 + face) * g.R + cw) * g.R + cu;
}

// Shadow rays of a wave through the light bins.  `cell` < 0: this lane has no ray for the bins.
// The wave serves one distinct cell at a time (rays of neighbouring pixels mostly share theirs).
// Lanes whose cell or face list overflowed are returned in the mask: they must walk the tree.
template <int MODE>
DEVINL unsigned long long light_bins_trace(const RenderParams &p, int cell, f3 o, f3 d, RaySt &s,
                                           int &n_tests, int &n_swept) {
  typedef unsigned long long mask_t;
  typedef const int32_t __attribute__((address_space(4))) *CI;
  const LightBins g = p.lbins;
  const SmemFetch<DevTri> tris{p.tri};
  const SmemFetch<DevSph> sphs{p.sph};
  const uint32_t nt = (uint32_t)p.n_tri;
  const int cells_per_face = g.R * g.R;
  mask_t todo = __builtin_amdgcn_ballot_w64(cell >= 0);
  mask_t fallback = 0;
  while (todo != 0) {
    const int lead = __builtin_ctzll(todo);
    const int c = __builtin_amdgcn_readlane(cell, lead);
    const mask_t same = __builtin_amdgcn_ballot_w64(cell == c) & todo;
    todo &= ~same;
    const bool mine = cell == c;
    const CI hdr = (CI)(uintptr_t)(g.face_hdr + (size_t)(c / cells_per_face) * kBinHdrInts);
    const CI cnt = (CI)(uintptr_t)(g.counts + 2 * (size_t)c);
    const int n_gt = hdr[0], n_gs = hdr[1], n_t = cnt[0], n_s = cnt[1];
    if (n_gt > kBinGlobalCap || n_gs > kBinGlobalCap || n_t > kBinCap || n_s > kBinCap) {
      fallback |= same;
      continue;
    }
    auto looking = [&]() { return (MODE == 1) ? (mine && s.key == kNoKey) : mine; };
    auto tri_list = [&](CI ids, int n) {
      for (int k = 0; k < n; ++k) {
        const bool act = looking();
        if (MODE == 1 && __builtin_amdgcn_ballot_w64(act) == 0) return;
        const int id = ids[k];
        const DevTri T[1] = {tris(id)};
        test_tris_general<MODE, 1>(T, [&](int) { return (uint32_t)id; }, o, d, s, act);
        n_tests += act ? 1 : 0;
        n_swept += 1;
      }
    };
    auto sph_list = [&](CI ids, int n) { // whole batches of 4: spare slots name valid spheres
      for (int k = 0; k < n; k += 4) {
        const bool act = looking();
        if (MODE == 1 && __builtin_amdgcn_ballot_w64(act) == 0) return;
        const int i0 = ids[k], i1 = ids[k + 1], i2 = ids[k + 2], i3 = ids[k + 3];
        const DevSph S[4] = {sphs(i0), sphs(i1), sphs(i2), sphs(i3)};
        test_sphs_general<MODE, 4>(
            S, [&](int i) { return nt + (uint32_t)(i == 0 ? i0 : i == 1 ? i1 : i == 2 ? i2 : i3); },
            o, d, s, act);
        n_tests += act ? 4 : 0;
        n_swept += 4;
      }
    };
    tri_list(hdr + 2, n_gt);
    tri_list((CI)(uintptr_t)(g.tri_ids + (size_t)c * kBinCap), n_t);
    if (p.n_sph > 0) {
      sph_list(hdr + 2 + kBinGlobalCap, (n_gs + 3) & ~3);
      sph_list((CI)(uintptr_t)(g.sph_ids + (size_t)c * kBinCap), (n_s + 3) & ~3);
    }
  }
  return fallback;
}

// ---------------------------------------------------------------------------------------
// The frame = two kernels on the same stream.
//
//   k_primary<STAGE, V, NV>  camera.h:31-34 get_ray + main.cpp:722 closest hit over every
//                            primitive; writes one 16-byte hit record per pixel.
//   k_shade<STAGE>           main.cpp:723-789: normal, per-light shadow ray (occlusion()) and
//                            Phong; fp32 RGB and/or the PPM-quantised bytes.
//
// One fused kernel was the first design (and is what "one work-item per pixel" suggests); it was
// split because the two halves want different things: the primary pass is fastest with 2 pixels
// per lane (every primitive fetch feeds 128 rays), the shadow pass with 1 (a wave retires as
// soon as its 64 rays are decided) plus re-packing, and fused they held so many values live
// across the hot loops that hipcc spilled SGPRs inside them.  The hand-over costs 133 MB of
// writes + reads per 4K frame (~0.05 ms) and one kernel boundary (~1.5 us).
//
// 256 threads = 4 waves; a wave covers (16*PX) x 4 pixels, a workgroup a (32*PX) x 8 tile.
// ---------------------------------------------------------------------------------------

// workgroup -> pixel tile.  The dispatcher deals blocks round-robin over the 8 XCDs (block b
// runs on XCD b % 8), so with the identity map every XCD gets every 8th tile of every image
// row: an even mix of cheap (sky: primary rays only) and expensive (floor: primary + shadow)
// tiles.  That balance is what matters here -- tiles share no data beyond the scene tables,
// which every XCD's L2 holds anyway.  (Giving each XCD one contiguous run of tiles, the usual
// GEMM remap, was measured first: the XCDs that drew sky rows went idle and the c4 frame took
// 23.6 ms instead of 18.3.)
template <int PX> struct Tile {
  int w0, lr0, h_tile; // first column, first local row, image row of local row lr0
  int lx0, ly;         // this lane: pixel q sits at column w0 + lx0 + 16 q, local row lr0 + ly
  int wave, lane;
  DEVINL Tile(const RenderParams &p) {
    constexpr int TW = 32 * PX;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int tid = threadIdx.x;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    lane = tid & 63;
    lx0 = (wave & 1) * (16 * PX) + (lane & 15);
    ly = ((wave >> 1) << 2) + (lane >> 4);
    w0 = tx * TW;
    // local row lr (ascending h) -> image row h.  A contiguous band has strip_rows >= its
    // height, so lr / strip_rows == 0 and h = h0 + lr; cyclic strips (multi-GPU) jump by
    // strip_step image rows per strip.  strip_rows is a multiple of kTileH (host-checked).
    lr0 = ty * kTileH;
    h_tile = p.h0 + (lr0 / p.strip_rows) * p.strip_step + (lr0 % p.strip_rows);
  }
};

// main.cpp:709-713 + camera.h:31-34
DEVINL f3 primary_dir(const RenderParams &p, int w, int h) {
  const f3 origin = mk(p.origin[0], p.origin[1], p.origin[2]);
  const float is = (float)w / (float)(p.W - 1);
  const float it = (float)h / (float)(p.H - 1);
  return normalize(((ld3(p.llc) + ld3(p.horizontal) * is) + ld3(p.vertical) * it) - origin);
}

template <int STAGE, typename V, int NV>
__global__ void __launch_bounds__(256) k_primary(const RenderParams p) {
  constexpr int PX = NV * lanes_of<V>::n; // pixels per work-item
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE == STAGE_LDS ? kLdsChunkBytes : 16];
  const Tile<PX> T(p);
  const int lr = T.lr0 + T.ly, h = T.h_tile + T.ly;
  const bool row_ok = (lr < p.n_local_rows) && (h < p.H);

  int w[PX];
  f3 dir[PX];
  Hit hit[PX];
#pragma unroll
  for (int q = 0; q < PX; ++q) {
    w[q] = T.w0 + T.lx0 + 16 * q;
    dir[q] = primary_dir(p, w[q], h);
    hit[q].t = FLT_MAX; // main.cpp:715
    hit[q].v = 0.f;
    hit[q].idx = -1;
  }

  // ---- main.cpp:722 closest hit over every primitive
  V3<V> dv[NV];
  pack3<V, NV>(dir, dv);
  if (STAGE == STAGE_SMEM) {
    closest_tri_primary<V, NV>(SmemFetch<DevTriP>{p.tri_p}, p.n_tri, 0, dv, hit);
    if constexpr (PX == 2) {
      // multiples of 8 through the hand-scheduled packed body, the tail through the generic one
      const int n8 = p.n_sph & ~7;
      closest_sph_primary_pk(SmemFetch<SphP2>{reinterpret_cast<const SphP2 *>(p.sph_p)}, n8,
                             p.n_tri, dv[0], hit);
      closest_sph_primary<V, NV>(SmemFetch<DevSphP>{p.sph_p + n8}, p.n_sph - n8, p.n_tri + n8, dv,
                                 hit);
    } else if constexpr (PX == 1) {
      closest_sph_primary_pairs(SmemFetch<PairP>{reinterpret_cast<const PairP *>(p.sph2_p)},
                                (p.n_sph + 1) >> 1, p.n_tri, dir[0], hit[0]);
    } else {
      closest_sph_primary<V, NV>(SmemFetch<DevSphP>{p.sph_p}, p.n_sph, p.n_tri, dv, hit);
    }
  } else {
    constexpr int CT = kLdsChunkBytes / (int)sizeof(DevTriP);
    for (int k0 = 0; k0 < p.n_tri; k0 += CT) {
      const int n = min(CT, p.n_tri - k0);
      __syncthreads();
      lds_stage(reinterpret_cast<DevTriP *>(lds_raw), p.tri_p + k0, n);
      __syncthreads();
      closest_tri_primary<V, NV>(LdsFetch<DevTriP>{reinterpret_cast<const DevTriP *>(lds_raw)},
                                 n, k0, dv, hit);
    }
    constexpr int CS = kLdsChunkBytes / (int)sizeof(DevSphP);
    for (int k0 = 0; k0 < p.n_sph; k0 += CS) {
      const int n = min(CS, p.n_sph - k0);
      __syncthreads();
      lds_stage(reinterpret_cast<DevSphP *>(lds_raw), p.sph_p + k0, n);
      __syncthreads();
      closest_sph_primary<V, NV>(LdsFetch<DevSphP>{reinterpret_cast<const DevSphP *>(lds_raw)},
                                 n, p.n_tri + k0, dv, hit);
    }
  }

  // ---- hand-over: t, v, idx per pixel (band-local pixel order), 16-byte stores
#pragma unroll
  for (int q = 0; q < PX; ++q)
    if (row_ok && w[q] < p.W) {
      HitRec r;
      r.t = hit[q].t;
      r.v = hit[q].v;
      r.idx = hit[q].idx;
      r.pad = 0;
      p.hits[(size_t)lr * p.W + w[q]] = r;
    }
}

#ifndef ESC_SHADE_WAVES
#define ESC_SHADE_WAVES 6
#endif
template <int STAGE>
__global__ void __launch_bounds__(256, ESC_SHADE_WAVES) k_shade(const RenderParams p) {
  typedef float V;
  constexpr int NV = 1;
  constexpr int TW = 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE == STAGE_LDS ? kLdsChunkBytes : 16];
  __shared__ float lds_px[TW * kTileH * 3];
  __shared__ RepackLds lds_rays; // SMEM stage only

  const Tile<1> T(p);
  const int tid = threadIdx.x;
  const int wave = T.wave, lane = T.lane;
  const int rows = p.n_local_rows;
  const int lr = T.lr0 + T.ly, h = T.h_tile + T.ly;
  const int w = T.w0 + T.lx0;
  const bool inside = (lr < rows) && (h < p.H) && (w < p.W);

  const f3 origin = mk(p.origin[0], p.origin[1], p.origin[2]);
  HitRec hr;
  hr.t = FLT_MAX;
  hr.v = 0.f;
  hr.idx = -1;
  hr.pad = 0;
  // ESC_STAGE_BVH is ONE kernel: the closest hit is found right here (screen bin of this tile,
  // else the tree walk) instead of being handed over through HBM by k_primary -- its search
  // holds few registers, so nothing spills, and the primary direction is computed once and kept
  // (two divides, a square root and three more divides each time otherwise).
  f3 dir_kept = mk(0.f, 0.f, 0.f);
  if constexpr (STAGE == STAGE_BVH) {
    if (inside) dir_kept = primary_dir(p, w, h); // camera.h:31-34
    RaySt s;
    s.tmax = inside ? FLT_MAX : 0.f; // main.cpp:715
    s.thit = 0.f;
    s.v = 0.f;
    s.key = kNoKey;
    int nv = 0, nt = 0, ns = 0;
    if (!bin_trace(p, T.w0 / 32, T.h_tile, dir_kept, s, inside))
      bvh_trace<0, true>(p, origin, dir_kept, s, inside, nv, nt, ns);
    hr.t = s.tmax;
    hr.v = s.v;
    hr.idx = (int32_t)s.key; // kNoKey -> -1
  } else {
    if (inside) hr = p.hits[(size_t)lr * p.W + w];
  }
  const bool has_hit = inside && (hr.idx >= 0);

  // ---- main.cpp:723-738 normal of the hit (per-lane gathers, once per pixel)
  f3 N = mk(0.f, 0.f, 0.f);
  int mi = 0;
  if (has_hit) {
    if (hr.idx < p.n_tri) {
      const DevTri Tr = p.tri[hr.idx];
      N = normalize(cross(ld3(Tr.e1), ld3(Tr.e2))); // :728-731
      mi = Tr.geom;
      if (p.mat[mi].has_normals) { // :733-738 with u == 0 (quirk S1)
        const DevTriN Q = p.tri_n[hr.idx];
        const float u = 0.f, v = hr.v;
        N = normalize((ld3(Q.n1) * u + ld3(Q.n2) * v) + ld3(Q.n0) * ((1.f - u) - v));
      }
    } else {
      const int k = hr.idx - p.n_tri;
      const DevSph S = p.sph[k];
      const f3 dir = (STAGE == STAGE_BVH) ? dir_kept : primary_dir(p, w, h);
      N = normalize((origin + dir * hr.t) - mk(S.cx, S.cy, S.cz)); // extension
      mi = p.sph_mat[k];
    }
  }

  // ---- main.cpp:740-789 per-light shading
  float t = hr.t;
  float r = 0.f, g = 0.f, b = 0.f; // vec3 default ctor, main.cpp:557-558
  const float nl = (float)p.n_lights;
  uint32_t n_shadow = 0;
  unsigned long long n_any = 0; // any-hit tests the reference would have executed
  int n_swept = 0;              // primitives this WAVE swept in any-hit loops (x64 = lane-tests)
  for (int li = 0; li < p.n_lights; ++li) {
    const DevLight Lt = p.lights[li];
    Any a[1];
    f3 ro = N, rL = N; // shadow-ray origin (main.cpp:757 `hit`) and unit direction
    f3 lP = N;         // the light sample point and its index (light bins, ESC_STAGE_BVH)
    int lpt = 0;
    a[0].tb = 0.f;
    a[0].tocc = 0.f;
    a[0].kocc = -1;
    if (has_hit) {
      // x % 1 == 0: a one-face light needs no draw (wave-uniform shortcut)
      const uint32_t face =
          (p.face_mode == 0) ? (uint32_t)p.fixed_face
          : (Lt.n_faces == 1) ? 0u
                              : face_hash(p.seed, (uint32_t)(h * p.W + w), (uint32_t)li,
                                          (uint32_t)Lt.n_faces);
      const f3 P = ld3(p.light_points + 4 * (Lt.first_point + (int)face)); // quirk S2
      lP = P;
      lpt = Lt.first_point + (int)face;
      // the primary direction is recomputed here (same ops, same bits) rather than kept in
      // registers across the any-hit loops of the previous light
      const f3 dir = (STAGE == STAGE_BVH) ? dir_kept : primary_dir(p, w, h);
      ro = origin + dir * (t - FLT_EPSILON); // :757-758
      rL = P - ro;                           // :759
      const float len = length(rL);          // :761
      t = len - FLT_EPSILON;                 // :764
      rL = normalize(rL);                    // :766
      a[0].tb = t;
    }
    if (!(a[0].tb > 0.f)) a[0].tb = 0.f; // dead rays carry tb = 0
    if (p.shadows) { // :772 occlusion(): wave-uniform loops
      if constexpr (STAGE == STAGE_BVH) {
        RaySt s;
        s.tmax = a[0].tb;
        s.thit = 0.f;
        s.v = 0.f;
        s.key = kNoKey;
        int n_visits = 0, n_tests = 0;
        // the occluder's t2 is only ever read by the NEXT light (quirk S3): the last light may
        // stop at any occluder, the others need the first one in primitive order
        const bool need_first = li + 1 < p.n_lights;
        const bool ray = a[0].tb > 0.f;
        bool walk = ray; // rays the light bins cannot serve walk the tree
        if (p.lbins.n_points > 0) {
          int cell = -1;
          if (ray && lpt < p.lbins.n_points) cell = light_cell(p.lbins, lpt, lP, ro);
          const unsigned long long fb =
              need_first ? light_bins_trace<2>(p, cell, ro, rL, s, n_tests, n_swept)
                         : light_bins_trace<1>(p, cell, ro, rL, s, n_tests, n_swept);
          walk = ray && (cell < 0 || ((fb >> lane) & 1ull) != 0);
        }
        if (need_first)
          bvh_trace<2, false>(p, ro, rL, s, walk, n_visits, n_tests, n_swept);
        else
          bvh_trace<1, false>(p, ro, rL, s, walk, n_visits, n_tests, n_swept);
        a[0].kocc = (int32_t)s.key; // kNoKey -> -1
        a[0].tocc = s.thit;
        n_any += (unsigned)n_tests;
      } else if constexpr (STAGE == STAGE_SMEM) {
        // ---- segments of the primitive list, undecided rays re-packed in between
        RepackLds &R = lds_rays;
        R.ox[tid] = ro.x; R.oy[tid] = ro.y; R.oz[tid] = ro.z;
        R.lx[tid] = rL.x; R.ly[tid] = rL.y; R.lz[tid] = rL.z;
        R.tb[tid] = a[0].tb;
        R.kocc[tid] = -1;
        R.tocc[tid] = 0.f;
        // Occluded rays mostly meet their occluder early in the list, so re-packing pays at
        // the beginning and not later: segment lengths double (256, 256, 512, 1024, ...
        // triangles; 512, 512, 1024, ... pair records), which keeps the barriers few.
        const int n_rec = (p.n_sph + 1) >> 1;
        int k0 = 0, seg = kSegTris; // triangles first (index order)
        bool in_tris = p.n_tri > 0;
        if (!in_tris) seg = kSegSphPairs;
        for (int sg = 0;; ++sg) {
          if (in_tris && k0 >= p.n_tri) {
            in_tris = false;
            k0 = 0;
            seg = kSegSphPairs;
            sg = 0;
          }
          if (!in_tris && k0 >= n_rec) break;
          const int n_here = min(seg, (in_tris ? p.n_tri : n_rec) - k0);
          const int n_live = repack_rays(R, tid);
          if (n_live == 0) break; // workgroup-uniform
          if (wave * 64 < n_live) { // otherwise this wave sits the segment out
            const int slot = wave * 64 + lane;
            const int rr = (slot < n_live) ? (int)R.list[slot] : -1;
            const int rs = (rr >= 0) ? rr : tid;
            Any aa[1];
            aa[0].tb = (rr >= 0) ? R.tb[rs] : 0.f;
            aa[0].tocc = 0.f;
            aa[0].kocc = -1;
            const f3 so = mk(R.ox[rs], R.oy[rs], R.oz[rs]);
            const f3 sL = mk(R.lx[rs], R.ly[rs], R.lz[rs]);
            if (in_tris) {
              const V3<V> sov[1] = {{so.x, so.y, so.z}}, sLv[1] = {{sL.x, sL.y, sL.z}};
              n_swept += n_here; // upper bound: exits inside a segment are not subtracted
              anyhit_tri<V, NV>(SmemFetch<DevTri>{p.tri + k0}, n_here, k0, sov, sLv, aa);
            } else {
              n_swept += 2 * anyhit_sph_pairs(
                                 SmemFetch<PairG>{reinterpret_cast<const PairG *>(p.sph2) + k0},
                                 n_here, p.n_tri + 2 * k0, so, sL, aa[0]);
            }
            if (aa[0].kocc >= 0) { // rr >= 0 here: a dead lane has tb = 0 and accepts nothing
              R.tb[rr] = 0.f;
              R.tocc[rr] = aa[0].tocc;
              R.kocc[rr] = aa[0].kocc;
            }
          }
          k0 += n_here;
          if (sg >= 1) seg *= 2;
        }
        __syncthreads();
        a[0].kocc = R.kocc[tid];
        a[0].tocc = R.tocc[tid];
        rL = mk(R.lx[tid], R.ly[tid], R.lz[tid]); // not kept live across the segments
      } else {
        const V3<V> ov[1] = {{ro.x, ro.y, ro.z}}, Lv[1] = {{rL.x, rL.y, rL.z}};
        constexpr int CT = kLdsChunkBytes / (int)sizeof(DevTri);
        for (int k0 = 0; k0 < p.n_tri; k0 += CT) {
          const int n = min(CT, p.n_tri - k0);
          __syncthreads();
          lds_stage(reinterpret_cast<DevTri *>(lds_raw), p.tri + k0, n);
          __syncthreads();
          anyhit_tri<V, NV>(LdsFetch<DevTri>{reinterpret_cast<const DevTri *>(lds_raw)}, n, k0,
                            ov, Lv, a);
        }
        constexpr int CS = kLdsChunkBytes / (int)sizeof(DevSph);
        for (int k0 = 0; k0 < p.n_sph; k0 += CS) {
          const int n = min(CS, p.n_sph - k0);
          __syncthreads();
          lds_stage(reinterpret_cast<DevSph *>(lds_raw), p.sph + k0, n);
          __syncthreads();
          anyhit_sph<V, NV>(LdsFetch<DevSph>{reinterpret_cast<const DevSph *>(lds_raw)}, n,
                            p.n_tri + k0, ov, Lv, a);
        }
      }
    }
    if (has_hit) {
      if (p.shadows) {
        n_shadow += 1u;
        // tests occlusion() runs for this ray: up to and including its first occluder
        if (STAGE != STAGE_BVH)
          n_any += (a[0].kocc >= 0) ? (unsigned)(a[0].kocc + 1) : (unsigned)(p.n_tri + p.n_sph);
      }
      if (p.shadows && a[0].kocc >= 0) {
        t = a[0].tocc; // occlusion() wrote the occluder's t2 through its reference (quirk S3)
      } else {         // :772-773 `continue` otherwise
        const float d = dot(N, rL); // :775
        if (!(d <= 0.f)) {          // :777
          const DevMat M = p.mat[mi];                  // :768
          // x / 1.0f == x bit for bit, so a single light skips the six correctly rounded divides
          f3 c = ld3(M.ka) * 0.5f + ld3(M.ke);         // :769-770
          if (nl != 1.f) c = c / nl;
          const f3 Hh = normalize((N + rL) * 2.f);     // :780
          const float sp = powf(dot(N, Hh), M.Ns);
          f3 ds = ld3(M.kd) * d + ld3(M.ks) * sp;      // :782-783
          if (nl != 1.f) ds = ds / nl;
          c = c + ds;
          r += c.x;                                       // :786-788
          g += c.y;
          b += c.z;
        }
      }
    }
  }

  // ---- counters: ballot + popcount per wave, summed per workgroup in LDS, then ONE global atomic
  // per counter per workgroup into one of kCounterSets replicas (each on its own cache line).
  // 130k waves adding to four words of one line took longer than shading itself (4.7 ms).
  if (p.counters) {
    __shared__ unsigned long long wg_cnt[5];
    if (tid < 5) wg_cnt[tid] = 0ull;
    __syncthreads();
    const uint32_t ni = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(inside));
    const uint32_t nh = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(has_hit));
    uint32_t ns = n_shadow;
    unsigned long long na = n_any;
    for (int o = 32; o > 0; o >>= 1) {
      ns += __shfl_down(ns, o);
      na += __shfl_down(na, o);
    }
    if (lane == 0) {
      atomicAdd(&wg_cnt[0], (unsigned long long)ni);
      atomicAdd(&wg_cnt[1], (unsigned long long)nh);
      atomicAdd(&wg_cnt[2], (unsigned long long)ns);
      atomicAdd(&wg_cnt[3], na);
      atomicAdd(&wg_cnt[4], (unsigned long long)n_swept * 64ull);
    }
    __syncthreads();
    if (tid < 5 && wg_cnt[tid])
      atomicAdd(&p.counters[(blockIdx.x % kCounterSets) * 8 + tid], wg_cnt[tid]);
  }

  // ---- framebuffer: transpose the tile through LDS so each store instruction writes
  // consecutive dwords of one image row (12-byte pixels would otherwise stride the lanes).
  const int lx = T.lx0, ly = T.ly, w0 = T.w0, lr0 = T.lr0;
  const bool full_tile = (w0 + TW <= p.W) && (lr0 + kTileH <= rows) && (T.h_tile + kTileH <= p.H);
  if (p.out_f32) {
    if (full_tile) {
      const int li = (ly * TW + lx) * 3;
      lds_px[li + 0] = r;
      lds_px[li + 1] = g;
      lds_px[li + 2] = b;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i; // 0 .. 767
        const int row = idx / (TW * 3), col = idx % (TW * 3);
        const size_t o = ((size_t)(lr0 + row) * p.W + w0) * 3 + col;
        p.out_f32[o] = lds_px[idx];
      }
    } else if (inside) {
      const size_t o = ((size_t)lr * p.W + w) * 3;
      p.out_f32[o + 0] = r;
      p.out_f32[o + 1] = g;
      p.out_f32[o + 2] = b;
    }
  }
  if (p.out_u8) { // main.cpp:676-682 clamp > 1, int(c * 255)
    const float cr = (r > 1.f) ? 1.f : r, cg = (g > 1.f) ? 1.f : g, cb = (b > 1.f) ? 1.f : b;
    const uint8_t qr = (uint8_t)(int)(cr * 255.f), qg = (uint8_t)(int)(cg * 255.f),
                  qb = (uint8_t)(int)(cb * 255.f);
    if (full_tile && (p.W & 3) == 0) {
      __syncthreads(); // lds_px reuse
      unsigned char *lb = reinterpret_cast<unsigned char *>(lds_px);
      const int li = (ly * TW + lx) * 3;
      lb[li + 0] = qr;
      lb[li + 1] = qg;
      lb[li + 2] = qb;
      __syncthreads();
      constexpr int ROW_DW = TW * 3 / 4; // dwords per tile row
      if (tid < ROW_DW * kTileH) {
        const int row = tid / ROW_DW, col = tid % ROW_DW;
        const size_t o = ((size_t)(lr0 + row) * p.W + w0) * 3 + (size_t)col * 4;
        *reinterpret_cast<uint32_t *>(p.out_u8 + o) = reinterpret_cast<const uint32_t *>(lb)[tid];
      }
    } else if (inside) {
      const size_t o = ((size_t)lr * p.W + w) * 3;
      p.out_u8[o + 0] = qr;
      p.out_u8[o + 1] = qg;
      p.out_u8[o + 2] = qb;
    }
  }
}

// ---------------------------------------------------------------------------------------
// multi-GPU: rank r rendered strips r, r+N, r+2N, ... (strip k = image rows [k*S, k*S+S)).
// After the gather, rank 0 holds N blocks of local rows; this kernel lays them out as one
// frame.  One thread per dword (or byte) of the frame; reads and writes are both row-contiguous.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_assemble_strips(const T *__restrict__ gathered, T *__restrict__ frame, size_t rank_pitch,
                  int n_ranks, int H, int strip_rows, int row_elems) {
  const int h = blockIdx.y;
  const int strip = h / strip_rows;
  const int rank = strip % n_ranks;
  const size_t local_row = (size_t)(strip / n_ranks) * strip_rows + (h % strip_rows);
  const T *src = gathered + (size_t)rank * rank_pitch + local_row * row_elems;
  T *dst = frame + (size_t)h * row_elems;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row_elems; i += gridDim.x * blockDim.x)
    dst[i] = src[i];
}

} // namespace esc

extern "C" int esc_launch_assemble(const void *gathered, void *frame, size_t rank_pitch_bytes,
                                   int n_ranks, int H, int strip_rows, size_t row_bytes,
                                   hipStream_t stream) {
  if (H <= 0 || row_bytes == 0) return 0;
  const bool dwords = (row_bytes % 4 == 0) && (rank_pitch_bytes % 4 == 0) &&
                      (((uintptr_t)gathered | (uintptr_t)frame) % 4 == 0);
  if (dwords) {
    const int n = (int)(row_bytes / 4);
    dim3 grid((unsigned)std::min((n + 255) / 256, 64), (unsigned)H);
    hipLaunchKernelGGL(esc::k_assemble_strips<uint32_t>, grid, dim3(256), 0, stream,
                       (const uint32_t *)gathered, (uint32_t *)frame, rank_pitch_bytes / 4, n_ranks,
                       H, strip_rows, n);
  } else {
    const int n = (int)row_bytes;
    dim3 grid((unsigned)std::min((n + 255) / 256, 64), (unsigned)H);
    hipLaunchKernelGGL(esc::k_assemble_strips<uint8_t>, grid, dim3(256), 0, stream,
                       (const uint8_t *)gathered, (uint8_t *)frame, rank_pitch_bytes, n_ranks, H,
                       strip_rows, n);
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// host-side launchers (called from rt_capi.cpp)
// ---------------------------------------------------------------------------------------
extern "C" int esc_launch_prepare(const esc::RenderParams *p, esc::DevTriP *tri_p,
                                  esc::DevSphP *sph_p, esc::DevSphPairP *sph2_p,
                                  hipStream_t stream) {
  const int n = p->n_tri > p->n_sph ? p->n_tri : p->n_sph;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(esc::k_prepare_primary, dim3((n + 255) / 256), dim3(256), 0, stream, p->tri,
                     tri_p, p->n_tri, p->sph, sph_p, sph2_p, p->n_sph, p->origin[0], p->origin[1],
                     p->origin[2]);
  return (int)hipGetLastError();
}

extern "C" int esc_launch_bin_primary(const esc::RenderParams *p, const esc::PrimBoxDev *tri_boxes,
                                      const esc::PrimBoxDev *sph_boxes, hipStream_t stream) {
  const int n = p->n_tri + p->n_sph;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(esc::k_bin_primary, dim3((n + 3) / 4), dim3(256), 0, stream, *p, tri_boxes,
                     sph_boxes); // one wave per primitive
  return (int)hipGetLastError();
}

extern "C" int esc_launch_bin_light(const esc::LightBins *g, const float *light_points,
                                    const esc::PrimBoxDev *tri_boxes, int n_tri,
                                    const esc::PrimBoxDev *sph_boxes, int n_sph,
                                    hipStream_t stream) {
  const int n = n_tri + n_sph;
  if (n <= 0 || g->n_points <= 0) return 0;
  hipLaunchKernelGGL(esc::k_bin_light, dim3((n + 3) / 4), dim3(256), 0, stream, *g, light_points,
                     tri_boxes, n_tri, sph_boxes, n_sph);
  return (int)hipGetLastError();
}

// hoisted leaf blocks for rays leaving (ox,oy,oz); n_* count block SLOTS (pads included)
extern "C" int esc_launch_prepare_bvh(const esc::DevTri *tri, esc::DevTriP *tri_p, int n_tri,
                                      const esc::DevSph *sph, esc::DevSphP *sph_p, int n_sph,
                                      float ox, float oy, float oz, hipStream_t stream) {
  const int n = std::max(n_tri, n_sph);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(esc::k_prepare_bvh, dim3((n + 255) / 256), dim3(256), 0, stream, tri, tri_p,
                     n_tri, sph, sph_p, n_sph, ox, oy, oz);
  return (int)hipGetLastError();
}

template <int STAGE, typename V, int NV>
static void launch_primary(const esc::RenderParams *p, hipStream_t stream) {
  const int tw = 32 * NV * esc::lanes_of<V>::n;
  const int tiles_x = (p->W + tw - 1) / tw;
  const int tiles_y = (p->n_local_rows + esc::kTileH - 1) / esc::kTileH;
  hipLaunchKernelGGL((esc::k_primary<STAGE, V, NV>), dim3(tiles_x * tiles_y), dim3(256), 0, stream, *p);
}

// stage: 1 SMEM, 2 LDS, 3 BVH (px ignored).  px: pixels per work-item of the primary pass (1, 2 or 4); the shade
// pass always carries one pixel per work-item.
extern "C" int esc_launch_render(const esc::RenderParams *p, int stage, int px, hipStream_t stream) {
  if (p->n_local_rows <= 0 || p->W <= 0) return 0;
  using esc::v2f;
  const int tiles_y = (p->n_local_rows + esc::kTileH - 1) / esc::kTileH;
  const int shade_grid = ((p->W + 31) / 32) * tiles_y;
  if (stage == esc::STAGE_BVH) { // one kernel: closest hit + shading
    hipLaunchKernelGGL((esc::k_shade<esc::STAGE_BVH>), dim3(shade_grid), dim3(256), 0, stream, *p);
  } else if (stage == esc::STAGE_LDS) {
    if (px == 1) launch_primary<esc::STAGE_LDS, float, 1>(p, stream);
    else if (px == 2) launch_primary<esc::STAGE_LDS, v2f, 1>(p, stream);
    else launch_primary<esc::STAGE_LDS, v2f, 2>(p, stream);
    hipLaunchKernelGGL((esc::k_shade<esc::STAGE_LDS>), dim3(shade_grid), dim3(256), 0, stream, *p);
  } else {
    if (px == 1) launch_primary<esc::STAGE_SMEM, float, 1>(p, stream);
    else if (px == 2) launch_primary<esc::STAGE_SMEM, v2f, 1>(p, stream);
    else launch_primary<esc::STAGE_SMEM, v2f, 2>(p, stream);
    hipLaunchKernelGGL((esc::k_shade<esc::STAGE_SMEM>), dim3(shade_grid), dim3(256), 0, stream, *p);
  }
  return (int)hipGetLastError();
}
