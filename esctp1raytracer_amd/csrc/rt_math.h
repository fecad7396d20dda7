// rt_math.h -- device-side vocabulary of the render kernels: vec.h arithmetic in reference order,
// lane vectors (packed fp32), the exact tails of the two primitive tests, and the staging
// front-ends (scalar-cache / LDS fetch of wave-uniform records).  Included by rt_kernels.hip only;
// everything here is subject to the arithmetic contract stated at the top of that file.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// staging front-ends
// ---------------------------------------------------------------------------------------

// SMEM: the table pointer and index are wave-uniform, so hipcc emits s_load_dwordx4/x8/x16
// and the VALU takes the values straight from SGPRs.
template <typename Rec> struct SmemFetch {
  const Rec *__restrict__ p;
  // Read through the CONSTANT address space: with a wave-uniform address hipcc then always
  // selects s_load, also behind barriers / fences, where its "is this global memory ever
  // written in the kernel?" analysis gives up and would fall back to per-lane global_load.
  // The tables are written before the launch and never by the frame kernels.
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

} // namespace esc
