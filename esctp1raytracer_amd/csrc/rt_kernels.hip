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

#include "rt_device.h"

namespace esc {

#define DEVINL __device__ __forceinline__

constexpr int STAGE_SMEM = 1;
constexpr int STAGE_LDS = 2;

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
// exact tails (rare paths)
// ---------------------------------------------------------------------------------------

// ray_triangle.h:21-54 given the fp32 numerators.  Returns true on accept.
DEVINL bool tri_exact(float detf, float unum, float vnum, float tnum, float tbound, float &t2o,
                      float &v2o) {
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
DEVINL bool sph_exact(float b, float disc, float tbound, float &t2o) {
  if (disc < 0.f) return false;
  float sq = sqrtf(disc);
  float t2 = -b - sq;
  if (t2 < FLT_EPSILON) t2 = -b + sq;
  if (t2 < FLT_EPSILON) return false;
  if (t2 >= tbound) return false;
  t2o = t2;
  return true;
}

// ---------------------------------------------------------------------------------------
// per-frame constants for primary rays
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_prepare_primary(const DevTri *__restrict__ tri, DevTriP *__restrict__ tri_p, int n_tri,
                  const DevSph *__restrict__ sph, DevSphP *__restrict__ sph_p, int n_sph,
                  float ox, float oy, float oz) {
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
  }
}

// ---------------------------------------------------------------------------------------
// primitive loops.  `rec(k)` yields record k wave-uniformly (SGPRs or LDS broadcast).
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
DEVINL void tri_primary_numerators(const DevTriP &T, f3 d, float &det, float &un, float &vn) {
  const f3 pv = cross(d, ld3(T.e2)); // ray_triangle.h:18
  det = dot(ld3(T.e1), pv);          // :21
  un = dot(ld3(T.tv), pv);           // :32 numerator
  vn = dot(d, ld3(T.qv));            // :40 numerator
}

DEVINL void test_tri2_primary(const DevTriP (&T)[2], int idx, f3 d, Hit &h) {
  float da, ua, va, db, ub, vb;
  tri_primary_numerators(T[0], d, da, ua, va);
  tri_primary_numerators(T[1], d, db, ub, vb);
  const bool ca = tri_candidate(da, ua, va), cb = tri_candidate(db, ub, vb);
  if (__builtin_amdgcn_ballot_w64(ca | cb)) { // wave-uniform skip of the f64 tail
    float t2, v2;
    if (ca && tri_exact(da, ua, va, T[0].tnum, h.t, t2, v2)) {
      h.t = t2;
      h.v = v2;
      h.idx = idx;
    }
    if (cb && tri_exact(db, ub, vb, T[1].tnum, h.t, t2, v2)) {
      h.t = t2;
      h.v = v2;
      h.idx = idx + 1;
    }
  }
}

template <typename Fetch>
DEVINL void closest_tri_primary(Fetch rec, int n, int base, f3 d, Hit &h) {
  const int n4 = n & ~3;
  if (n4) {
    DevTriP A[2], B[2];
    fetch_batch(rec, 0, A);
    for (int k = 0; k < n4; k += 4) {
      fetch_batch(rec, rec.landed(A[1].tnum, k + 2), B);
      test_tri2_primary(A, base + k, d, h);
      fetch_batch(rec, rec.landed(B[1].tnum, min(k + 4, n - 2)), A); // clamped: last one unused
      test_tri2_primary(B, base + k + 2, d, h);
    }
  }
  for (int k = n4; k < n; ++k) {
    const DevTriP T = rec(k);
    float da, ua, va, t2, v2;
    tri_primary_numerators(T, d, da, ua, va);
    if (tri_candidate(da, ua, va) && tri_exact(da, ua, va, T.tnum, h.t, t2, v2)) {
      h.t = t2;
      h.v = v2;
      h.idx = base + k;
    }
  }
}

// ---- closest hit, primary rays, spheres ---------------------------------------------------
DEVINL void test_sph4_primary(const DevSphP (&s)[4], int idx, f3 d, Hit &h) {
  float b[4], q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b[i] = (s[i].ocx * d.x + s[i].ocy * d.y) + s[i].ocz * d.z;
    q[i] = b[i] * b[i] - s[i].cc;
  }
  const float m = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3]));
  if (__builtin_amdgcn_ballot_w64(!(m < 0.f))) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t2;
      if (sph_exact(b[i], q[i], h.t, t2)) {
        h.t = t2;
        h.idx = idx + i;
      }
    }
  }
}

template <typename Fetch>
DEVINL void closest_sph_primary(Fetch rec, int n, int base, f3 d, Hit &h) {
  const int n8 = n & ~7;
  if (n8) {
    DevSphP A[4], B[4];
    fetch_batch(rec, 0, A);
    for (int k = 0; k < n8; k += 8) {
      fetch_batch(rec, rec.landed(A[3].cc, k + 4), B);
      test_sph4_primary(A, base + k, d, h);
      fetch_batch(rec, rec.landed(B[3].cc, min(k + 8, n - 4)), A);
      test_sph4_primary(B, base + k + 4, d, h);
    }
  }
  for (int k = n8; k < n; ++k) {
    const DevSphP s0 = rec(k);
    const float b0 = (s0.ocx * d.x + s0.ocy * d.y) + s0.ocz * d.z;
    const float q0 = b0 * b0 - s0.cc;
    float t2;
    if (sph_exact(b0, q0, h.t, t2)) {
      h.t = t2;
      h.idx = base + k;
    }
  }
}

// ---- any-hit (main.cpp:314-329), general origin -------------------------------------------
// Per-lane state of one occlusion() call.  tb is the bound: > 0 while the lane is still
// looking, set to 0 once it found its FIRST occluder (or if it never looked), so later
// primitives cannot accept (accepts need eps <= t2 < tb).  tocc receives that occluder's t2
// (occlusion() mutates the caller's t, quirk S3) and kocc its index in (triangles, spheres)
// order.  The wave leaves a loop early once no lane is looking (checked per block of
// kExitStride primitives, not per primitive).
struct Any {
  float tb;
  float tocc;
  int32_t kocc;
};
constexpr int kExitStride = 32;

DEVINL void test_tri_any(const DevTri &T, int idx, f3 o, f3 L, Any &a) {
  const f3 e1 = ld3(T.e1), e2 = ld3(T.e2);
  const f3 pv = cross(L, e2);    // ray_triangle.h:18
  const float det = dot(e1, pv); // :21
  const f3 tv = o - ld3(T.v0);   // :29
  const float un = dot(tv, pv);  // :32
  const f3 qv = cross(tv, e1);   // :37
  const float vn = dot(L, qv);   // :40
  const bool c = tri_candidate(det, un, vn);
  if (__builtin_amdgcn_ballot_w64(c)) {
    float t2, v2;
    if (c && tri_exact(det, un, vn, dot(e2, qv), a.tb, t2, v2)) {
      a.tocc = t2;
      a.kocc = idx;
      a.tb = 0.f;
    }
  }
}

template <typename Fetch>
DEVINL void anyhit_tri(Fetch rec, int n, int base, f3 o, f3 L, Any &a) {
  for (int k0 = 0; k0 < n; k0 += kExitStride) {
    if (!__builtin_amdgcn_ballot_w64(a.tb > 0.f)) return; // every lane done
    const int m = min(kExitStride, n - k0);
    const int m2 = m & ~1;
    if (m2) {
      DevTri A = rec(k0), B;
      for (int k = 0; k < m2; k += 2) {
        B = rec(rec.landed(A.e2[2], k0 + k + 1));
        __builtin_amdgcn_sched_barrier(0);
        test_tri_any(A, base + k0 + k, o, L, a);
        A = rec(rec.landed(B.e2[2], k0 + min(k + 2, m - 1)));
        __builtin_amdgcn_sched_barrier(0);
        test_tri_any(B, base + k0 + k + 1, o, L, a);
      }
    }
    if (m2 < m) test_tri_any(rec(k0 + m2), base + k0 + m2, o, L, a);
  }
}

DEVINL void accept_sph_any(float b, float q, int idx, Any &a) {
  float t2;
  if (sph_exact(b, q, a.tb, t2)) {
    a.tocc = t2;
    a.kocc = idx;
    a.tb = 0.f;
  }
}

DEVINL void test_sph2_any(const DevSph (&s)[2], int idx, f3 o, f3 L, Any &a) {
  float b[2], q[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f3 oc = mk(o.x - s[i].cx, o.y - s[i].cy, o.z - s[i].cz);
    b[i] = dot(oc, L);
    q[i] = b[i] * b[i] - (dot(oc, oc) - s[i].r2);
  }
  if (__builtin_amdgcn_ballot_w64(!(fmaxf(q[0], q[1]) < 0.f))) {
    accept_sph_any(b[0], q[0], idx, a);
    accept_sph_any(b[1], q[1], idx + 1, a);
  }
}

template <typename Fetch>
DEVINL void anyhit_sph(Fetch rec, int n, int base, f3 o, f3 L, Any &a) {
  for (int k0 = 0; k0 < n; k0 += kExitStride) {
    if (!__builtin_amdgcn_ballot_w64(a.tb > 0.f)) return;
    const int m = min(kExitStride, n - k0);
    const int m4 = m & ~3;
    if (m4) {
      DevSph A[2], B[2];
      fetch_batch(rec, k0, A);
      for (int k = 0; k < m4; k += 4) {
        fetch_batch(rec, rec.landed(A[1].r2, k0 + k + 2), B);
        test_sph2_any(A, base + k0 + k, o, L, a);
        fetch_batch(rec, rec.landed(B[1].r2, k0 + min(k + 4, m - 2)), A);
        test_sph2_any(B, base + k0 + k + 2, o, L, a);
      }
    }
    for (int k = m4; k < m; ++k) {
      const DevSph s0 = rec(k0 + k);
      const f3 oc = mk(o.x - s0.cx, o.y - s0.cy, o.z - s0.cz);
      const float b0 = dot(oc, L);
      const float q0 = b0 * b0 - (dot(oc, oc) - s0.r2);
      accept_sph_any(b0, q0, base + k0 + k, a);
    }
  }
}

// ---------------------------------------------------------------------------------------
// staging front-ends
// ---------------------------------------------------------------------------------------

// SMEM: the table pointer and index are wave-uniform, so hipcc emits s_load_dwordx4/x8/x16
// and the VALU takes the values straight from SGPRs.
template <typename Rec> struct SmemFetch {
  const Rec *__restrict__ p;
  DEVINL Rec operator()(int k) const { return p[k]; }
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
// the frame kernel
// ---------------------------------------------------------------------------------------
template <int STAGE>
__global__ void __launch_bounds__(256) k_render(const RenderParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE == STAGE_LDS ? kLdsChunkBytes : 16];
  __shared__ float lds_px[kTileW * kTileH * 3];

  // ---- workgroup -> pixel tile.  Blocks are dealt round-robin over the 8 XCDs, so block b
  // and b+8 share an L2; give each XCD one contiguous run of tiles (= contiguous framebuffer
  // rows) instead of every 8th tile.  Grid is padded to a multiple of 8; surplus blocks exit.
  const int rows = p.n_local_rows;
  const int tiles_x = (p.W + kTileW - 1) / kTileW;
  const int tiles_y = (rows + kTileH - 1) / kTileH;
  const int n_tiles = tiles_x * tiles_y;
  const int per_xcd = gridDim.x >> 3;
  const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= n_tiles) return; // whole workgroup leaves together (no barrier yet)
  const int tx = tile % tiles_x, ty = tile / tiles_x;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lx = ((wave & 1) << 4) + (lane & 15);
  const int ly = ((wave >> 1) << 2) + (lane >> 4);
  const int w = tx * kTileW + lx;
  // local row lr (ascending h) -> image row h.  A contiguous band has strip_rows >= its
  // height, so lr / strip_rows == 0 and h = h0 + lr; cyclic strips (multi-GPU) jump by
  // strip_step image rows per strip.  strip_rows is a multiple of kTileH (host-checked).
  const int lr0 = ty * kTileH;
  const int h_tile = p.h0 + (lr0 / p.strip_rows) * p.strip_step + (lr0 % p.strip_rows);
  const int lr = lr0 + ly;
  const int h = h_tile + ly;
  const bool inside = (w < p.W) && (lr < rows) && (h < p.H);

  // ---- main.cpp:709-713 + camera.h:31-34
  const f3 origin = mk(p.origin[0], p.origin[1], p.origin[2]);
  const float is = (float)w / (float)(p.W - 1);
  const float it = (float)h / (float)(p.H - 1);
  const f3 dir = normalize(((ld3(p.llc) + ld3(p.horizontal) * is) + ld3(p.vertical) * it) - origin);

  // ---- main.cpp:722 closest hit over every primitive
  Hit hit;
  hit.t = FLT_MAX;
  hit.v = 0.f;
  hit.idx = -1;
  if (STAGE == STAGE_SMEM) {
    closest_tri_primary(SmemFetch<DevTriP>{p.tri_p}, p.n_tri, 0, dir, hit);
    closest_sph_primary(SmemFetch<DevSphP>{p.sph_p}, p.n_sph, p.n_tri, dir, hit);
  } else {
    constexpr int CT = kLdsChunkBytes / (int)sizeof(DevTriP);
    for (int k0 = 0; k0 < p.n_tri; k0 += CT) {
      const int n = min(CT, p.n_tri - k0);
      __syncthreads();
      lds_stage(reinterpret_cast<DevTriP *>(lds_raw), p.tri_p + k0, n);
      __syncthreads();
      closest_tri_primary(LdsFetch<DevTriP>{reinterpret_cast<const DevTriP *>(lds_raw)}, n, k0,
                          dir, hit);
    }
    constexpr int CS = kLdsChunkBytes / (int)sizeof(DevSphP);
    for (int k0 = 0; k0 < p.n_sph; k0 += CS) {
      const int n = min(CS, p.n_sph - k0);
      __syncthreads();
      lds_stage(reinterpret_cast<DevSphP *>(lds_raw), p.sph_p + k0, n);
      __syncthreads();
      closest_sph_primary(LdsFetch<DevSphP>{reinterpret_cast<const DevSphP *>(lds_raw)}, n,
                          p.n_tri + k0, dir, hit);
    }
  }
  const bool has_hit = inside && (hit.idx >= 0);

  // ---- main.cpp:723-738 normal and material of the hit (per-lane gathers, once per pixel)
  f3 N = mk(0.f, 0.f, 0.f);
  f3 ka = N, kd = N, ks = N, ke = N;
  float Ns = 0.f;
  if (has_hit) {
    int mi;
    if (hit.idx < p.n_tri) {
      const DevTri T = p.tri[hit.idx];
      N = normalize(cross(ld3(T.e1), ld3(T.e2))); // :728-731
      mi = T.geom;
      if (p.mat[mi].has_normals) { // :733-738 with u == 0 (quirk S1)
        const DevTriN Q = p.tri_n[hit.idx];
        const float u = 0.f, v = hit.v;
        N = normalize((ld3(Q.n1) * u + ld3(Q.n2) * v) + ld3(Q.n0) * ((1.f - u) - v));
      }
    } else {
      const int k = hit.idx - p.n_tri;
      const DevSph S = p.sph[k];
      N = normalize((origin + dir * hit.t) - mk(S.cx, S.cy, S.cz)); // extension
      mi = p.sph_mat[k];
    }
    const DevMat M = p.mat[mi];
    ka = ld3(M.ka);
    kd = ld3(M.kd);
    ks = ld3(M.ks);
    ke = ld3(M.ke);
    Ns = M.Ns;
  }

  // ---- main.cpp:740-789 per-light shading
  float r = 0.f, g = 0.f, b = 0.f; // vec3 default ctor, main.cpp:557-558
  float t = hit.t;
  const float nl = (float)p.n_lights;
  uint32_t n_shadow = 0;
  unsigned long long n_any = 0; // any-hit tests the reference would have executed
  for (int li = 0; li < p.n_lights; ++li) {
    const DevLight Lt = p.lights[li];
    f3 hp = N, L = N;
    Any a;
    a.tb = 0.f;
    a.tocc = 0.f;
    a.kocc = -1;
    if (has_hit) {
      uint32_t face = (p.face_mode == 0)
                          ? (uint32_t)p.fixed_face
                          : face_hash(p.seed, (uint32_t)(h * p.W + w), (uint32_t)li,
                                      (uint32_t)Lt.n_faces);
      const f3 P = ld3(p.light_points + 4 * (Lt.first_point + (int)face)); // quirk S2
      hp = origin + dir * (t - FLT_EPSILON); // :757-758
      L = P - hp;                            // :759
      const float len = length(L);           // :761
      t = len - FLT_EPSILON;                 // :764
      L = normalize(L);                      // :766
      a.tb = t;
    }
    if (p.shadows) { // :772 occlusion(): wave-uniform loops, dead lanes carry tb = 0
      if (!(a.tb > 0.f)) a.tb = 0.f;
      if (STAGE == STAGE_SMEM) {
        if (__builtin_amdgcn_ballot_w64(a.tb > 0.f)) {
          anyhit_tri(SmemFetch<DevTri>{p.tri}, p.n_tri, 0, hp, L, a);
          anyhit_sph(SmemFetch<DevSph>{p.sph}, p.n_sph, p.n_tri, hp, L, a);
        }
      } else {
        constexpr int CT = kLdsChunkBytes / (int)sizeof(DevTri);
        for (int k0 = 0; k0 < p.n_tri; k0 += CT) {
          const int n = min(CT, p.n_tri - k0);
          __syncthreads();
          lds_stage(reinterpret_cast<DevTri *>(lds_raw), p.tri + k0, n);
          __syncthreads();
          anyhit_tri(LdsFetch<DevTri>{reinterpret_cast<const DevTri *>(lds_raw)}, n, k0, hp, L, a);
        }
        constexpr int CS = kLdsChunkBytes / (int)sizeof(DevSph);
        for (int k0 = 0; k0 < p.n_sph; k0 += CS) {
          const int n = min(CS, p.n_sph - k0);
          __syncthreads();
          lds_stage(reinterpret_cast<DevSph *>(lds_raw), p.sph + k0, n);
          __syncthreads();
          anyhit_sph(LdsFetch<DevSph>{reinterpret_cast<const DevSph *>(lds_raw)}, n,
                     p.n_tri + k0, hp, L, a);
        }
      }
      if (has_hit) {
        n_shadow += 1u;
        // tests occlusion() runs for this ray: up to and including its first occluder
        n_any += (a.kocc >= 0) ? (unsigned)(a.kocc + 1) : (unsigned)(p.n_tri + p.n_sph);
      }
    }
    if (has_hit) {
      f3 c = (ka * 0.5f + ke) / nl; // :769-770
      const bool occluded = p.shadows && (a.kocc >= 0);
      if (occluded) {
        t = a.tocc; // occlusion() wrote the occluder's t2 through its reference (quirk S3)
      } else {
        const float d = dot(N, L); // :775
        if (!(d <= 0.f)) {         // :777
          const f3 Hh = normalize((N + L) * 2.f); // :780
          const float sp = powf(dot(N, Hh), Ns);
          c = c + (kd * d + ks * sp) / nl; // :782-783
          r += c.x;                        // :786-788
          g += c.y;
          b += c.z;
        }
      }
    }
  }

  // ---- counters: one atomic per wave (ballot + popcount)
  if (p.counters) {
    const uint64_t mi = __builtin_amdgcn_ballot_w64(inside);
    const uint64_t mh = __builtin_amdgcn_ballot_w64(has_hit);
    uint32_t ns = n_shadow;
    unsigned long long na = n_any;
    for (int o = 32; o > 0; o >>= 1) {
      ns += __shfl_down(ns, o);
      na += __shfl_down(na, o);
    }
    if (lane == 0) {
      atomicAdd(&p.counters[0], (unsigned long long)__popcll(mi));
      atomicAdd(&p.counters[1], (unsigned long long)__popcll(mh));
      atomicAdd(&p.counters[2], (unsigned long long)ns);
      if (na) atomicAdd(&p.counters[3], na);
    }
  }

  // ---- framebuffer: transpose the 32x8 tile through LDS so each store instruction writes
  // consecutive dwords of one image row (12-byte pixels would otherwise stride the lanes).
  const int w0 = tx * kTileW;
  const bool full_tile = (w0 + kTileW <= p.W) && (lr0 + kTileH <= rows) && (h_tile + kTileH <= p.H);
  if (p.out_f32) {
    if (full_tile) {
      const int li = (ly * kTileW + lx) * 3;
      lds_px[li + 0] = r;
      lds_px[li + 1] = g;
      lds_px[li + 2] = b;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i; // 0..767
        const int row = idx / (kTileW * 3), col = idx % (kTileW * 3);
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
      const int li = (ly * kTileW + lx) * 3;
      lb[li + 0] = qr;
      lb[li + 1] = qg;
      lb[li + 2] = qb;
      __syncthreads();
      if (tid < kTileW * kTileH * 3 / 4) { // 192 dwords
        const int row = tid / (kTileW * 3 / 4), col = tid % (kTileW * 3 / 4);
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
                                  esc::DevSphP *sph_p, hipStream_t stream) {
  const int n = p->n_tri > p->n_sph ? p->n_tri : p->n_sph;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(esc::k_prepare_primary, dim3((n + 255) / 256), dim3(256), 0, stream, p->tri,
                     tri_p, p->n_tri, p->sph, sph_p, p->n_sph, p->origin[0], p->origin[1],
                     p->origin[2]);
  return (int)hipGetLastError();
}

extern "C" int esc_launch_render(const esc::RenderParams *p, int stage, hipStream_t stream) {
  const int rows = p->n_local_rows;
  if (rows <= 0 || p->W <= 0) return 0;
  const int tiles_x = (p->W + esc::kTileW - 1) / esc::kTileW;
  const int tiles_y = (rows + esc::kTileH - 1) / esc::kTileH;
  const int n_tiles = tiles_x * tiles_y;
  const int grid = ((n_tiles + 7) / 8) * 8;
  if (stage == esc::STAGE_LDS)
    hipLaunchKernelGGL(esc::k_render<esc::STAGE_LDS>, dim3(grid), dim3(256), 0, stream, *p);
  else
    hipLaunchKernelGGL(esc::k_render<esc::STAGE_SMEM>, dim3(grid), dim3(256), 0, stream, *p);
  return (int)hipGetLastError();
}
