// rt_accel.h -- ESC_STAGE_BVH: wave-synchronous tree walk, screen-space bins for
// primary rays, light-space bins for shadow rays, and the kernels that fill the bins.  Every
// structure only decides WHICH primitives get the exact tests of rt_math.h.  Included by
// rt_kernels.hip only.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"
#include "rt_math.h"

namespace esc {
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
// (kTinyTris: rt_device.h)
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
// PER-LANE lists: every lane reads its own cell's ids and records with vector loads (neighbouring
// lanes mostly share a cell, so the loads coalesce) and the wave runs until its longest list is
// done.  (Serving one distinct cell at a time from SGPRs, the way the tiles of the primary pass
// are served, was measured first: a wave's 64 rays land in ~2 cells on average, which halves the
// lane efficiency -- c4 0.299 vs 0.260 ms.)  Lanes whose cell or face list overflowed are
// returned in the mask: they must walk the tree.
template <int MODE>
DEVINL unsigned long long light_bins_trace(const RenderParams &p, int cell, f3 o, f3 d,
                                                RaySt &s, int &n_tests, int &n_swept) {
  const LightBins g = p.lbins;
  const uint32_t nt = (uint32_t)p.n_tri;
  const int cells_per_face = g.R * g.R;
  bool have = cell >= 0;
  const int32_t *hdr = g.face_hdr;
  int n_gt = 0, n_gs = 0, n_t = 0, n_s = 0;
  if (have) {
    hdr = g.face_hdr + (size_t)(cell / cells_per_face) * kBinHdrInts;
    n_gt = hdr[0];
    n_gs = hdr[1];
    n_t = g.counts[2 * (size_t)cell];
    n_s = g.counts[2 * (size_t)cell + 1];
  }
  const bool over = have && (n_gt > kBinGlobalCap || n_gs > kBinGlobalCap || n_t > kBinCap ||
                             n_s > kBinCap);
  const unsigned long long fallback = __builtin_amdgcn_ballot_w64(over);
  if (over) have = false;
  auto looking = [&]() { return (MODE == 1) ? (have && s.key == kNoKey) : have; };
  auto tri_list = [&](const int32_t *ids, int n) {
    for (int k = 0;; ++k) {
      const bool act = looking() && k < n;
      if (__builtin_amdgcn_ballot_w64(act) == 0) return;
      int id = 0;
      DevTri T[1];
      if (act) {
        id = ids[k];
        T[0] = p.tri[id];
      } else {
        T[0] = DevTri{};
      }
      test_tris_general<MODE, 1>(T, [&](int) { return (uint32_t)id; }, o, d, s, act);
      n_tests += act ? 1 : 0;
      n_swept += 1;
    }
  };
  auto sph_list = [&](const int32_t *ids, int n) { // spare slots of a batch name valid spheres
    for (int k = 0;; k += 4) {
      const bool act = looking() && k < n;
      if (__builtin_amdgcn_ballot_w64(act) == 0) return;
      int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
      DevSph S[4];
      if (act) {
        const int4 q = *reinterpret_cast<const int4 *>(ids + k);
        i0 = q.x; i1 = q.y; i2 = q.z; i3 = q.w;
        S[0] = p.sph[i0]; S[1] = p.sph[i1]; S[2] = p.sph[i2]; S[3] = p.sph[i3];
      } else {
        S[0] = S[1] = S[2] = S[3] = DevSph{0.f, 0.f, 0.f, -__builtin_huge_valf()};
      }
      test_sphs_general<MODE, 4>(
          S, [&](int i) { return nt + (uint32_t)(i == 0 ? i0 : i == 1 ? i1 : i == 2 ? i2 : i3); },
          o, d, s, act);
      n_tests += act ? 4 : 0;
      n_swept += 4;
    }
  };
  const size_t c = have ? (size_t)cell : 0;
  tri_list(hdr + 2, n_gt);
  tri_list(g.tri_ids + c * kBinCap, n_t);
  if (p.n_sph > 0) {
    sph_list(hdr + 2 + kBinGlobalCap, n_gs);
    sph_list(g.sph_ids + c * kBinCap, n_s);
  }
  return fallback;
}

} // namespace esc
