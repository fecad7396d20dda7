// rt_kernels.hip -- hand-written gfx950 kernels for the per-pixel render loop of
// pg42819/EscTp1RayTracer (reference paths relative to /root/reference):
//
//   k_prepare_primary / k_prepare_bvh   per-frame, per-primitive constants for rays leaving the
//                      camera origin
//   k_primary<STAGE>   camera.h:31-34 get_ray -> main.cpp:176-192 closest hit, brute force
//   k_shade<STAGE>     main.cpp:723-789 normal + per-light shadow ray (main.cpp:314-329) + Phong
//                      -> fp32 RGB and/or PPM-quantised bytes; under ESC_STAGE_BVH also the
//                      closest hit (one kernel per frame)
//   k_shadow_setup / k_anyhit_segment / k_shade_finish   the same shading pass in its queue form
//                      (long primitive lists on large bands): one launch per segment of the
//                      list, undecided rays compacted across the whole band in between
//   k_assemble_strips  multi-GPU: gathered strips -> frame
// The pieces live in rt_math.h (arithmetic vocabulary), rt_brute.h (brute-force loops and the
// conservative filters in front of them) and rt_accel.h (tree walk, screen / light bins and the
// kernels that fill them).
//
// Arithmetic contract: this file MUST be compiled with -ffp-contract=off (hipcc would
// otherwise fuse a*b+c into v_fma_f32 and flip pixels) and with
// correctly rounded fp32 divide/sqrt (hipcc default).  Every expression THAT REACHES THE IMAGE
// is evaluated in the order the reference evaluates it; the filters of rt_brute.h use explicit
// v_pk_fma_f32 and only decide whether that arithmetic runs for a pair.  The one liberty taken
// with the reference expressions themselves: vec.h:95-101 starts its dot
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
#include "rt_math.h"
#include "rt_brute.h"
#include "rt_lists.h"
#include "rt_accel.h"

namespace esc {

// ---------------------------------------------------------------------------------------
// per-frame constants for primary rays
// ---------------------------------------------------------------------------------------
// DevSphF of "the rays start oc away from a centre; cc = fl(fl-dot(oc, oc) - r2)": the scaled
// filter record with the margins of rt_brute.h "FILTERS" (r2a = |r2|)
DEVINL DevSphF sphere_filter_record(f3 oc, float cc, float r2a) {
  const float A = (fabsf(oc.x) + fabsf(oc.y)) + fabsf(oc.z);
  const float ccm = cc - ((A * A + r2a) * 0x1p-19f + 0x1p-120f);
  DevSphF F;
  F.sx = F.sy = F.sz = 0.f;
  F.w = 2.f; // always a candidate ...
  if (ccm > 0.f) {
    const float s = (sqrtf(ccm) * 0x1.fffff8p-1f - A * 0x1.2p-21f) * 0x1.fffff8p-1f;
    if (s > 0.f) { // ... unless the scaled test is well defined
      const float inv = 1.f / s;
      F.sx = oc.x * inv;
      F.sy = oc.y * inv;
      F.sz = oc.z * inv;
      F.w = 0.f;
    }
  }
  return F;
}

// per-frame forms of one triangle for rays leaving o: the exact hoisted record (DevTriP), the
// filter (DevTriF) and the pre-filter (DevTriPF) -- rt_brute.h
DEVINL void tri_primary_records(const DevTri &T, f3 o, DevTriP &Pout, DevTriF &Fout, DevTriPF &Qout) {
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
  Pout = P;
  { // filter form (rt_brute.h "Triangle FILTERS"): n1 = e2 x e1, n2 = e2 x tv, n3 = qv
    const f3 n1 = cross(e2, e1), n2 = cross(e2, tv);
    const float a1 = (fabsf(e1.x) + fabsf(e1.y)) + fabsf(e1.z);
    const float a2 = (fabsf(e2.x) + fabsf(e2.y)) + fabsf(e2.z);
    const float at = (fabsf(tv.x) + fabsf(tv.y)) + fabsf(tv.z);
    const float aq = (fabsf(qv.x) + fabsf(qv.y)) + fabsf(qv.z);
    const float p12 = a1 * a2;
    DevTriF F;
    F.n1[0] = n1.x; F.n1[1] = n1.y; F.n1[2] = n1.z;
    F.n2[0] = n2.x; F.n2[1] = n2.y; F.n2[2] = n2.z;
    F.n3[0] = qv.x; F.n3[1] = qv.y; F.n3[2] = qv.z;
    F.M = p12 * ((p12 + at * a2) + aq) * 0x1p-17f + 0x1p-120f;
    F.pad[0] = F.pad[1] = 0.f;
    Fout = F;
    // pre-filter form (rt_brute.h "Triangle pre-filter"): bounding sphere (G, R = 2 rho + slack)
    // seen from the ray origin o' = v0 + tv, and the normal scaled by 1 / tau'
    const f3 s3 = (e1 + e2) * (1.f / 3.f);
    const f3 ocg = tv - s3; // o' - G
    const float rho = sqrtf(fmaxf(fmaxf(dot(s3, s3), dot(e1 - s3, e1 - s3)), dot(e2 - s3, e2 - s3))) *
                      1.00001f;
    const float emax = sqrtf(fmaxf(dot(e1, e1), dot(e2, e2))) * 1.00001f;
    DevTriPF Q;
    Q.sx = Q.sy = Q.sz = 0.f;
    Q.w = 2.f; // always a candidate ...
    Q.gx = Q.gy = Q.gz = Q.pad = 0.f; // ... and always "grazing" (|0| <= 1)
    if (rho > 0x1p-10f * emax) { // ... unless the triangle is not a sliver
      // tau: |det| >= tau keeps the accepted hit within rho of the triangle; tau' adds what the
      // filter's own det can be off by
      const float tau = 0x1.99999ap+1f * 0x1p-24f * ((10.04f * at * a2 + 5.04f * at * a1) + 20.1f * p12) *
                        emax / rho;
      const float taup = (tau + 0x1.44p+3f * 0x1p-24f * p12) * 1.00001f + 0x1p-120f; // + 10.1u P12
      const float R = 2.f * rho + 0x1p-21f * ((at + a1) + a2); // + 8u (|tv| + |e1| + |e2|)
      const float A = (fabsf(ocg.x) + fabsf(ocg.y)) + fabsf(ocg.z);
      const float R2 = R * R * 1.00001f;
      const float ccg = dot(ocg, ocg) - R2;
      const float ccm = ccg - ((A * A + R2) * 0x1p-19f + 0x1p-120f);
      if (ccm > 0.f) {
        const float sc = (sqrtf(ccm) * 0x1.fffff8p-1f - A * 0x1.2p-21f) * 0x1.fffff8p-1f;
        if (sc > 0.f) {
          const float inv = 1.f / sc;
          Q.sx = ocg.x * inv;
          Q.sy = ocg.y * inv;
          Q.sz = ocg.z * inv;
          Q.w = 0.f;
        }
      }
      const float ig = 1.f / taup;
      Q.gx = n1.x * ig;
      Q.gy = n1.y * ig;
      Q.gz = n1.z * ig;
    }
    Qout = Q;
  }
}

__global__ void __launch_bounds__(256)
k_prepare_primary(const DevTri *__restrict__ tri, DevTriP *__restrict__ tri_p,
                  DevTriF *__restrict__ tri_f, DevTriPF *__restrict__ tri_pf, int n_tri,
                  const DevSph *__restrict__ sph, DevSphP *__restrict__ sph_p,
                  DevSphF *__restrict__ sph_f, int n_sph, float ox, float oy, float oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const f3 o = mk(ox, oy, oz);
  if (i < n_tri) {
    tri_primary_records(tri[i], o, tri_p[i], tri_f[i], tri_pf[i]);
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
    sph_f[i] = sphere_filter_record(oc, P.cc, fabsf(S.r2)); // rt_brute.h "FILTERS"
  }
}

// per-frame records of the sphere groups (rt_device.h SphGroups): the sorted spheres' primary and
// filter forms, and every group's bounding sphere in filter form.  R = rgeo + what the reference's
// rounding can add to a member's reach (rt_brute.h "Sphere groups").
__global__ void __launch_bounds__(256) k_prepare_groups(const SphGroups g, float ox, float oy, float oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const f3 o = mk(ox, oy, oz);
  if (i < g.n_grp * kSphGroup) {
    const DevSph S = g.sorted[i];
    DevSphP P;
    DevSphF F;
    if (S.r2 == -__builtin_huge_valf()) { // pad slot: disc = -inf, b'' = 0
      P.ocx = P.ocy = P.ocz = 0.f;
      P.cc = __builtin_huge_valf();
      F.sx = F.sy = F.sz = F.w = 0.f;
    } else {
      const f3 oc = o - mk(S.cx, S.cy, S.cz);
      P.ocx = oc.x;
      P.ocy = oc.y;
      P.ocz = oc.z;
      P.cc = dot(oc, oc) - S.r2;
      F = sphere_filter_record(oc, P.cc, fabsf(S.r2));
    }
    g.sorted_p[i] = P;
    g.sorted_f[i] = F;
  }
  if (i < g.n_grp + g.n_sup + g.n_hyp) { // groups, super-groups, hyper-groups: same form
    const DevSphGroup G = g.grp[i];
    DevSphF F;
    F.sx = F.sy = F.sz = F.w = 0.f; // pad group: never a candidate
    if (!(G.rgeo < 0.f)) {
      const f3 oc = o - mk(G.cx, G.cy, G.cz);
      const float A = (fabsf(oc.x) + fabsf(oc.y)) + fabsf(oc.z);
      const float R = (G.rgeo + 0x1.2p-10f * (A + 2.f * G.rgeo)) + 0x1p-60f;
      const float R2 = R * R * 1.00001f;
      F = sphere_filter_record(oc, dot(oc, oc) - R2, R2);
    }
    g.grp_f[i] = F;
  }
}

// per-frame records of the triangle groups (rt_device.h TriGroups): the sorted triangles' forms,
// and every group's / super-group's / hyper-group's record in pre-filter form (rt_brute.h
// "Triangle GROUPS").  Three launches: level 0 builds each group's cone of THIS frame from its 8
// members (statement (P): only members whose plane the camera is within H_t of can be accepted
// through the escape, and only by rays with |d . n_t| < beta_t); levels 1 and 2 merge their
// children's cones: for a child cone (a_c, s_c, beta_c) -- every flagged member t below it has
// |d . a_c| <= beta_t + |d| s_c -- and a parent axis a_p, |d . a_p| <= beta_t + |d| (s_c + |a_p x a_c|).
DEVINL DevTriPF tri_group_record(const DevTriGroup &G, f3 o, const DevTriEsc &E) {
  DevTriPF Q;
  Q.sx = Q.sy = Q.sz = Q.w = 0.f; // pad group: never within reach ...
  Q.gx = 0x1p60f;                 // ... and never "nearly parallel" (|d.x| <= 2^-60 opens pads: harmless)
  Q.gy = Q.gz = Q.pad = 0.f;
  if (G.rgeo < 0.f) return Q;
  Q.w = 2.f; // always open ...
  Q.gx = 0.f;
  if (G.always != 0.f) return Q; // ... unless the static bounds are usable
  const f3 oc = o - mk(G.cx, G.cy, G.cz);
  const float A = (fabsf(oc.x) + fabsf(oc.y)) + fabsf(oc.z);
  const float at = A + G.rext; // >= |tvec_t|_1 for every member
  const float R = (G.rgeo + 0x1p-21f * at) + 0x1p-60f;
  const float R2 = R * R * 1.00001f;
  const DevSphF F = sphere_filter_record(oc, dot(oc, oc) - R2, R2);
  Q.sx = F.sx;
  Q.sy = F.sy;
  Q.sz = F.sz;
  Q.w = F.w;
  if (E.state == 0) {
    Q.gx = 0x1p60f; // no ray of this frame can take the escape here
  } else if (E.state == 1) {
    // 1e-5: the fp32 normals and axes; 2^-20: the FMA chain of g'' and |d| - 1
    const float kp = (((E.s + 1e-5f) * 1.0001f + E.beta) + 0x1p-20f) * 1.0001f;
    if (kp < 1.f) {
      const float ik = 1.f / kp;
      Q.gx = E.ax * ik;
      Q.gy = E.ay * ik;
      Q.gz = E.az * ik;
    }
  } // state 2: g'' = 0, always "nearly parallel"
  return Q;
}

// the cone of THIS frame over the members [first, first + kTriGroup) that can take the escape at
// threshold tau / slack_k
DEVINL DevTriEsc tri_group_cone(const DevTri *sorted, int first, f3 o, float slack_k) {
  DevTriEsc N;
  N.ax = N.ay = N.az = N.s = N.beta = 0.f;
  N.state = 0;
  f3 acc = mk(0.f, 0.f, 0.f), ref = mk(0.f, 0.f, 0.f);
  int cnt = 0;
  for (int m = 0; m < kTriGroup; ++m) {
    const TriEscape E = tri_escape(sorted[first + m], o, slack_k);
    if (!E.possible) continue;
    if (!E.bounded) {
      N.state = 2;
      return N;
    }
    if (cnt == 0) ref = E.nh;
    acc = acc + E.nh * ((dot(E.nh, ref) < 0.f) ? -1.f : 1.f);
    N.beta = fmaxf(N.beta, E.beta);
    ++cnt;
  }
  if (cnt == 0) return N;
  const float an = sqrtf(dot(acc, acc));
  if (!(an > 0.5f * (float)cnt)) {
    N.state = 2; // no useful axis
    return N;
  }
  const f3 ax = acc * (1.f / an);
  float smax = 0.f;
  for (int m = 0; m < kTriGroup; ++m) {
    const TriEscape E = tri_escape(sorted[first + m], o, slack_k);
    if (!E.possible) continue;
    const f3 c = cross(ax, E.nh);
    smax = fmaxf(smax, sqrtf(dot(c, c)));
  }
  N.ax = ax.x;
  N.ay = ax.y;
  N.az = ax.z;
  N.s = smax;
  N.state = 1;
  return N;
}

__global__ void __launch_bounds__(256) k_prepare_tri_groups(const TriGroups g, float ox, float oy, float oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const f3 o = mk(ox, oy, oz);
  if (i < g.n_grp * kTriGroup)
    tri_primary_records(g.sorted[i], o, g.sorted_p[i], g.sorted_f[i], g.sorted_pf[i]);
  if (i < g.n_grp) {
    const int n_nodes = g.n_grp + g.n_sup + g.n_hyp;
    const DevTriEsc N0 = tri_group_cone(g.sorted, i * kTriGroup, o, g.grp[i].slack); // its own k
    g.esc[i] = N0;                                                                    // chain 0
    // chains 1 and 2 at the slack factors of this group's super- and hyper-group
    const float k_sup = g.grp[g.n_grp + i / kTriSuper].slack;
    const float k_hyp = g.grp[g.n_grp + g.n_sup + i / (kTriSuper * kTriHyper)].slack;
    g.esc[n_nodes + i] = tri_group_cone(g.sorted, i * kTriGroup, o, k_sup);
    g.esc[2 * n_nodes + i] = tri_group_cone(g.sorted, i * kTriGroup, o, k_hyp);
    g.grp_pf[i] = tri_group_record(g.grp[i], o, N0);
  }
}

// level 1 (super-groups from groups; chains 1 and 2) and level 2 (hyper-groups from super-groups;
// chain 2): one thread per (node, chain)
__global__ void __launch_bounds__(256) k_prepare_tri_merge(const TriGroups g, int level, float ox, float oy,
                                                           float oz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = level == 1 ? g.n_sup : g.n_hyp;
  const int n_chain = level == 1 ? 2 : 1; // chains that pass through this level
  if (t >= n * n_chain) return;
  const int j = t % n, chain = 3 - n_chain + t / n; // level 1: chains 1, 2; level 2: chain 2
  const int n_nodes = g.n_grp + g.n_sup + g.n_hyp;
  DevTriEsc *esc = g.esc + (size_t)chain * n_nodes;
  const int fan = level == 1 ? kTriSuper : kTriHyper;
  const int child0 = (level == 1 ? 0 : g.n_grp) + j * fan; // children are consecutive nodes
  const int self = (level == 1 ? g.n_grp : g.n_grp + g.n_sup) + j;
  DevTriEsc N;
  N.ax = N.ay = N.az = N.s = N.beta = 0.f;
  N.state = 0;
  f3 acc = mk(0.f, 0.f, 0.f), ref = mk(0.f, 0.f, 0.f);
  int cnt = 0;
  for (int c = 0; c < fan; ++c) {
    const DevTriEsc E = esc[child0 + c];
    if (E.state == 0) continue;
    if (E.state == 2) {
      N.state = 2;
      break;
    }
    const f3 a = mk(E.ax, E.ay, E.az);
    if (cnt == 0) ref = a;
    acc = acc + a * ((dot(a, ref) < 0.f) ? -1.f : 1.f);
    N.beta = fmaxf(N.beta, E.beta);
    ++cnt;
  }
  if (N.state != 2 && cnt > 0) {
    const float an = sqrtf(dot(acc, acc));
    if (!(an > 0.5f * (float)cnt)) {
      N.state = 2;
    } else {
      const f3 ax = acc * (1.f / an);
      float smax = 0.f;
      for (int c = 0; c < fan; ++c) {
        const DevTriEsc E = esc[child0 + c];
        if (E.state != 1) continue;
        const f3 x = cross(ax, mk(E.ax, E.ay, E.az));
        smax = fmaxf(smax, (sqrtf(dot(x, x)) + E.s) * 1.0001f + 1e-6f);
      }
      N.ax = ax.x;
      N.ay = ax.y;
      N.az = ax.z;
      N.s = smax;
      N.state = smax < 1.f ? 1 : 2;
    }
  }
  esc[self] = N;
  // the level's own record takes its own chain: super-groups chain 1, hyper-groups chain 2
  if (chain == level) g.grp_pf[self] = tri_group_record(g.grp[self], mk(ox, oy, oz), N);
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
// The frame = two kernels on the same stream.
//
//   k_primary<STAGE, V, NV>  camera.h:31-34 get_ray + main.cpp:722 closest hit over every
//                            primitive; writes the hit planes (idx for every pixel, t / v for hits).
//   k_shade<STAGE>           main.cpp:723-789: normal, per-light shadow ray (occlusion()) and
//                            Phong; fp32 RGB and/or the PPM-quantised bytes.
//
// One fused kernel was the first design (and is what "one work-item per pixel" suggests); it was
// split because the two halves want different things: the primary pass is fastest with 2 pixels
// per lane (every primitive fetch feeds 128 rays), the shadow pass with 1 (a wave retires as
// soon as its 64 rays are decided) plus re-packing, and fused they held so many values live
// across the hot loops that hipcc spilled SGPRs inside them.  The hand-over (hit planes: an index
// per pixel, t -- and v when normals exist -- per hit pixel) costs ~55 MB of writes + reads per 4K
// frame and one kernel boundary.
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
    init(p, blockIdx.x % tiles_x, blockIdx.x / tiles_x, threadIdx.x);
    wave = __builtin_amdgcn_readfirstlane(wave); // wave-uniform for the workgroup's own thread id
  }
  // a kernel launched on a (tiles_x, tiles_y) grid: no division for the tile's coordinates (an integer
  // division is ~25 instructions here, and k_frame asks for its tile six times over -- see tile_again2)
  struct Grid2D {};
  DEVINL Tile(const RenderParams &p, Grid2D) {
    init(p, (int)blockIdx.x, (int)blockIdx.y, threadIdx.x);
    wave = __builtin_amdgcn_readfirstlane(wave);
  }
  // tile (tx, ty) of the band as seen by thread `tid` of a 256-thread workgroup
  DEVINL Tile(const RenderParams &p, int tx, int ty, int tid) { init(p, tx, ty, tid); }
  DEVINL void init(const RenderParams &p, int tx, int ty, int tid) {
    constexpr int TW = 32 * PX;
    wave = tid >> 6; // `tid` may be another thread's id (per lane): no readfirstlane here
    lane = tid & 63;
    lx0 = (wave & 1) * (16 * PX) + (lane & 15);
    ly = ((wave >> 1) << 2) + (lane >> 4);
    w0 = tx * TW;
    // local row lr (ascending h) -> image row h.  A contiguous band has strip_rows >= its
    // height, so lr / strip_rows == 0 and h = h0 + lr; cyclic strips (multi-GPU) jump by
    // strip_step image rows per strip.  strip_rows is a multiple of kTileH (host-checked).
    lr0 = ty * kTileH;
    // (the two usual cases without a division: one contiguous band; strips of exactly one tile row)
    if (lr0 < p.strip_rows) h_tile = p.h0 + lr0;
    else if (p.strip_rows == kTileH) h_tile = p.h0 + ty * p.strip_step;
    else h_tile = p.h0 + (lr0 / p.strip_rows) * p.strip_step + (lr0 % p.strip_rows);
  }
};

// ---- counters: ballot + popcount per wave, summed per workgroup in LDS, then ONE global atomic
// per counter per workgroup into one of kCounterSets replicas (each on its own cache line).
// 130k waves adding to four words of one line took longer than shading itself (4.7 ms).
// Called by all 256 threads (barriers inside).  count_pixels: also add primary rays / hit pixels.
// sum of a per-lane count over the wave, as a wave-uniform (SGPR) 64-bit value
DEVINL unsigned long long wave_sum(uint32_t v) {
  unsigned long long s = v;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)s);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(s >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

DEVINL void emit_counters(const RenderParams &p, int tid, int lane, bool inside, bool has_hit,
                          uint32_t n_shadow, unsigned long long n_any,
                          unsigned long long lane_tests_wave, bool count_pixels) {
  if (!p.counters) return;
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
    if (count_pixels) {
      atomicAdd(&wg_cnt[0], (unsigned long long)ni);
      atomicAdd(&wg_cnt[1], (unsigned long long)nh);
    }
    atomicAdd(&wg_cnt[2], (unsigned long long)ns);
    atomicAdd(&wg_cnt[3], na);
    atomicAdd(&wg_cnt[4], lane_tests_wave);
  }
  __syncthreads();
  if (tid < 5 && wg_cnt[tid])
    atomicAdd(&p.counters[(blockIdx.x % kCounterSets) * 8 + tid], wg_cnt[tid]);
}

// ---- framebuffer: transpose the tile through LDS so each store instruction writes
// consecutive dwords of one image row (12-byte pixels would otherwise stride the lanes).
// lds_px: 32 * kTileH * 3 floats.  Called by all 256 threads.
DEVINL void write_tile(const RenderParams &p, const Tile<1> &T, int tid, float r, float g, float b,
                       bool inside, float *lds_px) {
  constexpr int TW = 32;
  const int rows = p.n_local_rows;
  const int lr = T.lr0 + T.ly, w = T.w0 + T.lx0;
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

// main.cpp:709-713 + camera.h:31-34
DEVINL f3 primary_dir(const RenderParams &p, int w, int h) {
  const f3 origin = mk(p.origin[0], p.origin[1], p.origin[2]);
  const float is = (float)w / (float)(p.W - 1);
  const float it = (float)h / (float)(p.H - 1);
  return normalize(((ld3(p.llc) + ld3(p.horizontal) * is) + ld3(p.vertical) * it) - origin);
}

// main.cpp:768-788 for one light whose shadow ray found no occluder: this light's Phong term added
// to (r, g, b).  nl = the number of lights as a float.
// The specular term ks * pow(dot(N, H), Ns) costs a normalise and a powf; it is skipped (a branch the
// whole wave takes together on the BASELINE scenes) where it is +-0 whatever the power is: the
// material is flagged spec_free (rt_device.h material_spec_free: ks == +-0, Ns in [0, 1024]) AND the
// power is certain to be finite and >= +0.  That holds when |N|^2 and |L|^2, evaluated right here, lie in
// [0.999, 1.001]: both are then finite, d = fl-dot(N, L) > 0 is not a NaN and the real N.L >= -10u, so
// S = N + L has |S|^2 >= 1.99, 2S normalises without underflow or overflow to a unit H, and
// x = fl-dot(N, H) lies within 12u of (|N|^2 + N.L) / |S|, in [0.49, 1.001]; powf of such a base to
// an exponent in [0, 1024] is a finite value in [+0, 2.8].  ks * that is ks itself, bit for bit
// (+-0 times a finite non-negative value keeps the sign of the zero), which is what `sp = 1` gives.
DEVINL void phong_add(const DevMat &M, f3 N, f3 rL, float nl, float &r, float &g, float &b) {
  const float d = dot(N, rL); // :775
  if (!(d <= 0.f)) {          // :777
    // x / 1.0f == x bit for bit, so a single light skips the six correctly rounded divides
    f3 c = ld3(M.ka) * 0.5f + ld3(M.ke); // :769-770
    if (nl != 1.f) c = c / nl;
    const float nn = dot(N, N), ll = dot(rL, rL);
    const bool skip = M.spec_free != 0 && nn >= 0.999f && nn <= 1.001f && ll >= 0.999f && ll <= 1.001f;
    float sp = 1.f;
    if (!skip) {
      const f3 Hh = normalize((N + rL) * 2.f); // :780
      sp = powf(dot(N, Hh), M.Ns);
    }
    f3 ds = ld3(M.kd) * d + ld3(M.ks) * sp; // :782-783
    if (nl != 1.f) ds = ds / nl;
    c = c + ds;
    r += c.x; // :786-788
    g += c.y;
    b += c.z;
  }
}

// This thread's tile position, RECOMPUTED from the thread id through an opaque copy: used by
// k_shade<SMEM> inside and after the light loop, so that pixel coordinates and tile fields do not
// occupy VGPRs across the any-hit sweeps (they cost 36 B of scratch per lane there otherwise).
DEVINL Tile<1> tile_again(const RenderParams &p) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int tiles_x = (p.W + 31) / 32;
  Tile<1> T(p, blockIdx.x % tiles_x, blockIdx.x / tiles_x, t);
  T.wave = __builtin_amdgcn_readfirstlane(T.wave);
  return T;
}

// GRP: the frame sweeps primitive groups (rt_device.h SphGroups / TriGroups).  The variant without
// them (small scenes, ESC_RENDER_INDEX_ORDER, ESC_RENDER_EXACT_ONLY) is its own instantiation so
// that the linear loops keep their registers (with both in one kernel the linear triangle loop of
// c5 went from 427 to 507 ms).
// main.cpp:722 for the PX pixels of every lane: the closest hit over every primitive, through the
// tile lists, the group sweeps or the linear loops (k_primary and the fused k_frame share it)
template <int STAGE, typename V, int NV, bool GRP, int PX>
DEVINL void primary_closest(const RenderParams &p, const Tile<PX> &T, const V3<V> (&dv)[NV], Hit (&hit)[PX],
                            unsigned char *lds_raw) {
  // this wave's 32 x 4 pixel tile in the band (rt_device.h TileLists); a wave wholly outside the
  // band has nothing to test
  const int tile_x = (T.w0 >> 5) + (T.wave & 1), tile_y = (T.lr0 >> 2) + (T.wave >> 1);
  const int tiles_x = (p.W + 31) >> 5;
  const bool tile_ok = tile_x < tiles_x && tile_y * 4 < p.n_local_rows;
  const int tile = tile_y * tiles_x + tile_x;
  if (STAGE == STAGE_SMEM) {
    if constexpr (PX == 2) {
      if (GRP && p.use_filter && p.tg.n_grp > 0) {
        const SmemFetch<TriPF> recp{reinterpret_cast<const TriPF *>(p.tg.sorted_pf)};
        const SmemFetch<TriF> recf{reinterpret_cast<const TriF *>(p.tg.sorted_f)};
        const SmemFetch<DevTriP> rece{p.tg.sorted_p};
        const SmemFetch<DevIdx4> reci{p.tg.orig};
        bool listed = false;
        if (p.tl.enabled) // this tile's triangles straight away (rt_lists.h)
          listed = !tile_ok ||
                   sweep_tile_list(p.tl, tile, [&](int i0, int i1, int i2, int i3) {
                     tri2_listed_primary(recf, rece, reinterpret_cast<const int32_t *>(p.tg.orig), i0, i1, dv[0], hit);
                     tri2_listed_primary(recf, rece, reinterpret_cast<const int32_t *>(p.tg.orig), i2, i3, dv[0], hit);
                   });
        if (!listed)
          closest_tri_primary_groups(
              SmemFetch<TriPF>{reinterpret_cast<const TriPF *>(p.tg.grp_pf) + p.tg.n_grp + p.tg.n_sup},
              SmemFetch<TriPF>{reinterpret_cast<const TriPF *>(p.tg.grp_pf) + p.tg.n_grp},
              SmemFetch<TriPF>{reinterpret_cast<const TriPF *>(p.tg.grp_pf)}, recp, recf, rece, reci,
              p.tg.n_hyp, dv[0], hit);
      } else {
      const int n2 = (p.use_filter && p.n_tri >= 8) ? (p.n_tri & ~3) : 0;
      closest_tri_primary_filter(SmemFetch<TriPF>{reinterpret_cast<const TriPF *>(p.tri_pf)},
                                 SmemFetch<TriF>{reinterpret_cast<const TriF *>(p.tri_f)},
                                 SmemFetch<DevTriP>{p.tri_p}, n2, 0, dv[0], hit);
      closest_tri_primary<V, NV>(SmemFetch<DevTriP>{p.tri_p + n2}, p.n_tri - n2, n2, dv, hit);
      }
    } else {
      closest_tri_primary<V, NV>(SmemFetch<DevTriP>{p.tri_p}, p.n_tri, 0, dv, hit);
    }
    if constexpr (PX == 2) {
      // multiples of 8 through the hand-scheduled packed bodies, the tail through the generic one
      if (GRP && p.use_filter && p.sg.n_grp > 0) {
        const SmemFetch<SphF2> recf{reinterpret_cast<const SphF2 *>(p.sg.sorted_f)};
        const SmemFetch<DevSphP> rece{p.sg.sorted_p};
        const SmemFetch<DevIdx4> reci{p.sg.orig};
        bool listed = false;
        if (p.sl.enabled)
          listed = !tile_ok ||
                   sweep_tile_list(p.sl, tile, [&](int i0, int i1, int i2, int i3) {
                     sph4_listed_primary(rece, reinterpret_cast<const int32_t *>(p.sg.orig), i0, i1, i2, i3,
                                         p.n_tri, dv[0], hit);
                   });
        if (!listed)
          closest_sph_primary_groups(
              SmemFetch<SphF2>{reinterpret_cast<const SphF2 *>(p.sg.grp_f) + p.sg.n_grp + p.sg.n_sup},
              SmemFetch<SphF2>{reinterpret_cast<const SphF2 *>(p.sg.grp_f) + p.sg.n_grp},
              SmemFetch<SphF2>{reinterpret_cast<const SphF2 *>(p.sg.grp_f)}, recf, rece, reci, p.sg.n_hyp,
              p.n_tri, dv[0], hit);
      } else {
      const int n8 = p.n_sph & ~7;
      if (p.use_filter)
        closest_sph_primary_filter(SmemFetch<SphF2>{reinterpret_cast<const SphF2 *>(p.sph_f)},
                                   SmemFetch<DevSphP>{p.sph_p}, n8, p.n_tri, dv[0], hit);
      else
        closest_sph_primary_pk(SmemFetch<SphP2>{reinterpret_cast<const SphP2 *>(p.sph_p)}, n8,
                               p.n_tri, dv[0], hit);
      closest_sph_primary<V, NV>(SmemFetch<DevSphP>{p.sph_p + n8}, p.n_sph - n8, p.n_tri + n8, dv,
                                 hit);
      }
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

}

template <int STAGE, typename V, int NV, bool GRP = false>
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
  primary_closest<STAGE, V, NV, GRP, PX>(p, T, dv, hit, lds_raw);

  // ---- hand-over: idx for every pixel, t (and v when normals exist) for hit pixels; plane
  // stores, band-local pixel order
#pragma unroll
  for (int q = 0; q < PX; ++q)
    if (row_ok && w[q] < p.W) {
      const size_t px = (size_t)lr * p.W + w[q];
      p.hits.idx[px] = hit[q].idx;
      if (hit[q].idx >= 0) {
        p.hits.t[px] = hit[q].t;
        if (p.tri_n) p.hits.v[px] = hit[q].v;
      }
    }
}

// waves per SIMD asked of the register allocator: 6 (80 VGPRs) measured best for the SMEM and BVH
// variants (5 and 4 were 1 % slower / no different); the LDS variant's 46 KB of LDS allow 4
// TGRP: the scene has triangle groups (rt_device.h TriGroups).  Their sweep needs more registers
// than 6 waves per SIMD leave (36 B of scratch per lane otherwise), so that variant is built for 5
// and scenes without triangle groups keep the leaner kernel.
constexpr int kListServed = 0x40000000;    // RepackLds::n_open: this ray's sphere phase ran on light lists
constexpr int kTriListServed = 0x20000000; // ... its triangle phase did
// ---------------------------------------------------------------------------------------
// The shadow pass of ONE light for the 256 rays of a workgroup, primitives through the scalar cache
// (main.cpp:772 occlusion(), wave-uniform loops): used by k_shade<SMEM> and by the fused k_frame.
// In: every thread's ray (ro, rL, a0.tb; tb = 0: no ray).  Out: a0.kocc / a0.tocc (the occluder in
// (triangles, spheres) order and its t2, quirk S3), what the sweep was (for the counters) and the
// per-pixel shading state, which is parked in LDS while the sweeps run so that it holds no VGPRs
// across them.  All 256 threads must call it (barriers inside).
// ---------------------------------------------------------------------------------------
template <bool TGRP>
DEVINL void shadow_sweep_smem(const RenderParams &p, RepackLds &R, int tid, int wave, int lane, int li, f3 ro,
                              f3 &rL, Any (&a)[1], f3 &N, float &r, float &g, float &b, float &t, int &mi,
                              int &grp_open, bool &tri_groups, bool &sph_groups, int &n_swept) {
  typedef float V;
  constexpr int NV = 1;
  // ---- segments of the primitive list, undecided rays re-packed in between
  // (the LAST light sweeps the spheres by decreasing solid angle when the host built that
  // order: its occluder is never read again -- rt_device.h sph2_ord)
  const bool ord = (li == p.n_lights - 1) && p.sph2_ord != nullptr;
  const DevSphPair *sweep_e = ord ? p.sph2_ord : p.sph2;
  const DevSphPairF *sweep_f = ord ? p.sph2_f_ord : p.sph2_f;
    R.ox[tid] = ro.x; R.oy[tid] = ro.y; R.oz[tid] = ro.z;
  R.lx[tid] = rL.x; R.ly[tid] = rL.y; R.lz[tid] = rL.z;
  R.tb[tid] = a[0].tb;
  R.kocc[tid] = -1;
  R.tocc[tid] = 0.f;
  R.n_open[tid] = 0;
  R.keep[0][tid] = N.x; R.keep[1][tid] = N.y; R.keep[2][tid] = N.z;
  R.keep[3][tid] = r; R.keep[4][tid] = g; R.keep[5][tid] = b;
  R.keep[6][tid] = t; R.keep[7][tid] = __int_as_float(mi);
  // Occluded rays mostly meet their occluder early in the list, so re-packing pays at
  // the beginning and not later: segment lengths double (256, 256, 512, 1024, ...
  // triangles; 512, 512, 1024, ... pair records), which keeps the barriers few.
  // the last light sweeps the sphere GROUPS when the host built them (rt_device.h SphGroups):
  // k0 then counts pair records of the sorted table, 4 per group, segments whole steps
  // Lights before the last sweep the groups too, in "first occluder" mode (rt_brute.h Any:
  // every group visited, the accepted primitive with the lowest original index kept) --
  // as long as the whole table is one segment, which it is below 2^20 records.
  const bool last_light = li == p.n_lights - 1;
  const bool grp = p.use_filter && p.sg.n_grp > 0 &&
                   (last_light || p.sg.n_grp * (kSphGroup / 2) <= kSegGroupPairs);
  const int n_rec = grp ? p.sg.n_grp * (kSphGroup / 2) : (p.n_sph + 1) >> 1;
  // ... and the triangle GROUPS (rt_device.h TriGroups): k0 counts sorted slots, 8 per group
  const bool tgrp = TGRP && p.use_filter && p.tg.n_grp > 0 &&
                    (last_light || p.tg.n_grp * kTriGroup <= kSegGroupPairs);
  const int n_tri_sweep = tgrp ? p.tg.n_grp * kTriGroup : p.n_tri;
  int k0 = 0, seg = tgrp ? kSegGroupPairs : kSegTris; // triangles first (index order)
  bool in_tris = p.n_tri > 0;
  const int seg_sph = grp ? kSegGroupPairs : kSegSphPairs;
  if (!in_tris) seg = seg_sph;
  for (int sg = 0;; ++sg) {
    if (in_tris && k0 >= n_tri_sweep) {
      in_tris = false;
      k0 = 0;
      seg = seg_sph;
      sg = 0;
    }
    if (!in_tris && k0 >= n_rec) break;
    const int n_here = min(seg, (in_tris ? n_tri_sweep : n_rec) - k0);
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
      aa[0].orig = nullptr;
      aa[0].orig_bias = aa[0].orig_add = 0;
      const f3 so = mk(R.ox[rs], R.oy[rs], R.oz[rs]);
      const f3 sL = mk(R.lx[rs], R.ly[rs], R.lz[rs]);
      if (TGRP && in_tris && tgrp) {
        bool far;
        const RayTF rt = make_ray_tri_filter(so, sL, p.shadow_center, p.shadow_rho_max, far);
        const RayF rs = make_ray_filter(so, sL, p.shadow_center);
        int n_open = 0;
        if (!last_light) aa[0].orig = reinterpret_cast<const int32_t *>(p.tg.orig); // bias, add 0
        // a light with one sample point: its triangle light lists (rt_lists.h) -- unless a ray starts
        // outside the region they were built for, or its cell overflowed
        bool served = false;
        const bool outside = far || !(so.x >= p.scene_lo[0] && so.x <= p.scene_hi[0] && so.y >= p.scene_lo[1] &&
                                      so.y <= p.scene_hi[1] && so.z >= p.scene_lo[2] && so.z <= p.scene_hi[2]);
        if (p.lt.enabled && li < p.lt.n_listed && n_tri_sweep <= kSegGroupPairs &&
            __builtin_amdgcn_ballot_w64(aa[0].tb > 0.f && outside) == 0) {
          const f3 Pl = ld3(p.light_points + 4 * p.lt.point[li]);
          int n_tests = 0, sw = 0;
          served = anyhit_tri_light_lists(
              p.lt, light_list_cell(p.lt, li, Pl, so),
              SmemFetch<TriPairF>{reinterpret_cast<const TriPairF *>(p.tg.sorted2_f)},
              SmemFetch<DevTri>{p.tg.sorted}, 0, so, sL, rt, aa, n_tests, sw);
          n_swept += sw;
          if (rr >= 0) R.n_open[rr] = (R.n_open[rr] + (n_tests >> 3)) | (served ? kTriListServed : 0);
        }
        if (!served) {
        // k0 sorted slots in = k0 / 8 groups = k0 / kPerSup super-groups = k0 / kPerHyp hyper-groups;
        // two per pair record
        constexpr int kPerSup = kTriGroup * kTriSuper, kPerHyp = kPerSup * kTriHyper;
        n_swept += anyhit_tri_groups_filter(
            SmemFetch<TriPairPF>{reinterpret_cast<const TriPairPF *>(p.tg.grp2_pf) +
                                 ((p.tg.n_grp + p.tg.n_sup) >> 1) + k0 / (2 * kPerHyp)},
            SmemFetch<TriPairPF>{reinterpret_cast<const TriPairPF *>(p.tg.grp2_pf) +
                                 (p.tg.n_grp >> 1) + k0 / (2 * kPerSup)},
            SmemFetch<TriPairPF>{reinterpret_cast<const TriPairPF *>(p.tg.grp2_pf) + (k0 >> 4)},
            SmemFetch<TriPairPF>{reinterpret_cast<const TriPairPF *>(p.tg.sorted2_pf) + (k0 >> 1)},
            SmemFetch<TriPairF>{reinterpret_cast<const TriPairF *>(p.tg.sorted2_f) + (k0 >> 1)},
            SmemFetch<DevTri>{p.tg.sorted + k0}, n_here / kPerHyp, k0, so, sL, rs,
            rt, far, aa,
            n_open);
        if (rr >= 0 && n_open) R.n_open[rr] += n_open;
        }
      } else if (in_tris) {
        const V3<V> sov[1] = {{so.x, so.y, so.z}}, sLv[1] = {{sL.x, sL.y, sL.z}};
        n_swept += n_here; // upper bound: exits inside a segment are not subtracted
        if (p.use_filter && n_here >= 8) { // k0 is even: segment lengths are
          bool far;
          const RayTF rt = make_ray_tri_filter(so, sL, p.shadow_center, p.shadow_rho_max, far);
          const RayF rs = make_ray_filter(so, sL, p.shadow_center);
          anyhit_tri_filter(
              SmemFetch<TriPairPF>{reinterpret_cast<const TriPairPF *>(p.tri2_pf) + (k0 >> 1)},
              SmemFetch<TriPairF>{reinterpret_cast<const TriPairF *>(p.tri2_f) + (k0 >> 1)},
              SmemFetch<DevTri>{p.tri + k0}, n_here, k0, so, sL, rs, rt, far, aa);
        } else {
          anyhit_tri<V, NV>(SmemFetch<DevTri>{p.tri + k0}, n_here, k0, sov, sLv, aa);
        }
      } else if (grp) {
        const RayF rf = make_ray_filter(so, sL, p.shadow_center);
        const float a1 = (fabsf(so.x - p.shadow_center[0]) + fabsf(so.y - p.shadow_center[1])) +
                         fabsf(so.z - p.shadow_center[2]);
        const bool far = !(a1 <= p.shadow_rho_max); // also catches NaN
        int n_open = 0;
        if (!last_light) {
          aa[0].orig = reinterpret_cast<const int32_t *>(p.sg.orig);
          aa[0].orig_bias = aa[0].orig_add = p.n_tri;
        }
        // a light with one sample point this frame: the cells of its light lists (rt_lists.h)
        // hold every sphere a ray's line can reach -- unless a ray starts outside the region
        // the reach was computed for, or its cell overflowed: then the sweep below runs
        bool served = false;
        const bool outside = !(so.x >= p.scene_lo[0] && so.x <= p.scene_hi[0] && so.y >= p.scene_lo[1] &&
                               so.y <= p.scene_hi[1] && so.z >= p.scene_lo[2] && so.z <= p.scene_hi[2]);
        if (p.ll.enabled && li < p.ll.n_listed && n_rec <= kSegGroupPairs &&
            __builtin_amdgcn_ballot_w64(aa[0].tb > 0.f && outside) == 0) {
          const f3 Pl = ld3(p.light_points + 4 * p.ll.point[li]);
          int n_tests = 0, sw = 0;
          served = anyhit_sph_light_lists(
              p.ll, light_list_cell(p.ll, li, Pl, so),
              SmemFetch<PairG>{reinterpret_cast<const PairG *>(p.sg.sorted2)}, p.n_tri, so, sL, aa[0],
              n_tests, sw);
          n_swept += sw;
          if (rr >= 0) R.n_open[rr] = (R.n_open[rr] + (n_tests >> 3)) | (served ? kListServed : 0);
        }
        if (!served)
        // k0 pair records in = k0 / 4 groups = k0 / 32 super-groups = k0 / 256 hyper-groups; two per record
        n_swept += anyhit_sph_groups_filter(
            SmemFetch<PairF>{reinterpret_cast<const PairF *>(p.sg.grp2_f) +
                             ((p.sg.n_grp + p.sg.n_sup) >> 1) + (k0 >> 9)},
            SmemFetch<PairF>{reinterpret_cast<const PairF *>(p.sg.grp2_f) + (p.sg.n_grp >> 1) + (k0 >> 6)},
            SmemFetch<PairF>{reinterpret_cast<const PairF *>(p.sg.grp2_f) + (k0 >> 3)},
            SmemFetch<PairF>{reinterpret_cast<const PairF *>(p.sg.sorted2_f) + k0},
            SmemFetch<PairG>{reinterpret_cast<const PairG *>(p.sg.sorted2) + k0}, n_here >> 8,
            p.n_tri + 2 * k0, so, sL, rf, far, aa[0], n_open);
        if (rr >= 0 && n_open) R.n_open[rr] += n_open;
      } else if (p.use_filter) {
        const RayF rf = make_ray_filter(so, sL, p.shadow_center);
        n_swept += 2 * anyhit_sph_pairs_filter(
                           SmemFetch<PairF>{reinterpret_cast<const PairF *>(sweep_f) + k0},
                           SmemFetch<PairG>{reinterpret_cast<const PairG *>(sweep_e) + k0},
                           n_here, p.n_tri + 2 * k0, so, sL, rf, aa[0]);
      } else {
        n_swept += 2 * anyhit_sph_pairs(
                           SmemFetch<PairG>{reinterpret_cast<const PairG *>(sweep_e) + k0},
                           n_here, p.n_tri + 2 * k0, so, sL, aa[0]);
      }
      if (aa[0].kocc >= 0) { // rr >= 0 here: a dead lane has tb = 0 and accepts nothing
        R.tb[rr] = 0.f;
        R.tocc[rr] = aa[0].tocc;
        R.kocc[rr] = aa[0].kocc;
      }
    }
    k0 += n_here;
    if (sg >= 1 && seg < (1 << 29)) seg *= 2;
  }
  __syncthreads();
  a[0].kocc = R.kocc[tid];
  a[0].tocc = R.tocc[tid];
  if (last_light) { // earlier lights report the reference's own count: their kocc is the
    if (grp || tgrp) grp_open = R.n_open[tid]; // first occluder in index order
    tri_groups = tgrp;
    sph_groups = grp;
  }
  rL = mk(R.lx[tid], R.ly[tid], R.lz[tid]); // not kept live across the segments
  N = mk(R.keep[0][tid], R.keep[1][tid], R.keep[2][tid]);
  r = R.keep[3][tid]; g = R.keep[4][tid]; b = R.keep[5][tid];
  t = R.keep[6][tid];
  mi = __float_as_int(R.keep[7][tid]);
}

// "Did any wave of the workgroup raise its hand?" with ONE barrier (__syncthreads_or takes three and an
// LDS reduction).  Four LDS words used in turn: vote k writes word k & 3 before the barrier and reads
// it after; on its way in every wave clears the NEXT word, (k + 1) & 3 -- last read before barrier
// k - 2 ... k - 1 at the latest, and written again only after barrier k.  `word` must be zero before
// the first vote (wg_vote_init + one barrier at the top of the kernel); `mine` is wave-uniform.
DEVINL void wg_vote_init(int *word) {
  if (threadIdx.x < 4) word[threadIdx.x] = 0;
  __syncthreads();
}
DEVINL bool wg_vote_any(int *word, int &k, bool mine) {
  const int cur = k & 3;
  ++k;
  if ((threadIdx.x & 63u) == 0) {
    word[(cur + 1) & 3] = 0;
    if (mine) word[cur] = 1;
  }
  __syncthreads();
  return word[cur] != 0;
}

// ---- the shadow rays of light li when LISTS serve them: no re-packing, no LDS, each wave on its own.
// shadow_sweep_smem re-packs the workgroup's undecided rays between segments of long sweeps; with the
// light lists (rt_lists.h) a sweep is a few batches, and the parking / re-packing / barriers around
// it cost more than the tests (c4: 0.20 of a 0.345 ms frame).  Launch-uniform condition: every
// primitive kind of the scene is either short enough to be tested directly (< 8 triangles: the same
// anyhit_tri call the segment loop makes) or has light lists for this light.  The calls, the
// bookkeeping of `Any` between the kinds (triangles first; a ray a triangle stopped does not look
// at the spheres) and the counter flags are those of shadow_sweep_smem, statement for statement.
template <bool TGRP> DEVINL bool shadow_lists_cover(const RenderParams &p, int li) {
  if (!p.use_filter) return false;
  bool tri_ok = p.n_tri == 0 || p.n_tri < 8;
  if (TGRP && p.tg.n_grp > 0)
    tri_ok = p.lt.enabled && li < p.lt.n_listed && p.tg.n_grp * kTriGroup <= kSegGroupPairs;
  const bool sph_ok = p.n_sph == 0 || (p.sg.n_grp > 0 && p.ll.enabled && li < p.ll.n_listed &&
                                       p.sg.n_grp * (kSphGroup / 2) <= kSegGroupPairs);
  return tri_ok && sph_ok;
}
// false: some live ray of this wave cannot be served (it starts outside the region the lists were
// built for, or its cell overflowed) -- the caller runs shadow_sweep_smem for the workgroup with
// the untouched `a`.  true: a.kocc / a.tocc hold what the sweep would have reported.
template <bool TGRP>
DEVINL bool shadow_wave_lists(const RenderParams &p, int li, f3 so, f3 sL, Any &a, int &grp_open,
                              bool &tri_groups, bool &sph_groups, int &n_swept) {
  const bool last_light = li == p.n_lights - 1;
  const bool tgrp = TGRP && p.tg.n_grp > 0;
  if (__builtin_amdgcn_ballot_w64(a.tb > 0.f) == 0) { // no ray in this wave (sky, lights behind): nothing
    if (last_light) {                                  // to sweep; the flags of a sweep that tested nothing
      const bool grp = p.sg.n_grp > 0;
      if (grp || tgrp) grp_open = 0;
      tri_groups = tgrp;
      sph_groups = grp;
    }
    return true;
  }
  int n_open = 0, sw_all = 0;
  Any out = a;
  float tb = a.tb;
  const bool in_box = so.x >= p.scene_lo[0] && so.x <= p.scene_hi[0] && so.y >= p.scene_lo[1] &&
                      so.y <= p.scene_hi[1] && so.z >= p.scene_lo[2] && so.z <= p.scene_hi[2];
  if (p.n_tri > 0) {
    Any aa[1];
    aa[0].tb = tb;
    aa[0].tocc = 0.f;
    aa[0].kocc = -1;
    aa[0].orig = nullptr;
    aa[0].orig_bias = aa[0].orig_add = 0;
    if (tgrp) {
      bool far;
      const RayTF rt = make_ray_tri_filter(so, sL, p.shadow_center, p.shadow_rho_max, far);
      if (!last_light) aa[0].orig = reinterpret_cast<const int32_t *>(p.tg.orig);
      if (__builtin_amdgcn_ballot_w64(tb > 0.f && (far || !in_box)) != 0) return false;
      const f3 Pl = ld3(p.light_points + 4 * p.lt.point[li]);
      int n_tests = 0, sw = 0;
      if (!anyhit_tri_light_lists(p.lt, light_list_cell(p.lt, li, Pl, so),
                                  SmemFetch<TriPairF>{reinterpret_cast<const TriPairF *>(p.tg.sorted2_f)},
                                  SmemFetch<DevTri>{p.tg.sorted}, 0, so, sL, rt, aa, n_tests, sw))
        return false;
      sw_all += sw;
      if (tb > 0.f) n_open = (n_open + (n_tests >> 3)) | kTriListServed;
    } else if (__builtin_amdgcn_ballot_w64(tb > 0.f) != 0) {
      const V3<float> sov[1] = {{so.x, so.y, so.z}}, sLv[1] = {{sL.x, sL.y, sL.z}};
      sw_all += p.n_tri;
      anyhit_tri<float, 1>(SmemFetch<DevTri>{p.tri}, p.n_tri, 0, sov, sLv, aa);
    }
    if (aa[0].kocc >= 0) {
      tb = 0.f;
      out.tocc = aa[0].tocc;
      out.kocc = aa[0].kocc;
    }
  }
  if (p.n_sph > 0) {
    Any aa;
    aa.tb = tb;
    aa.tocc = 0.f;
    aa.kocc = -1;
    aa.orig = nullptr;
    aa.orig_bias = aa.orig_add = 0;
    if (!last_light) {
      aa.orig = reinterpret_cast<const int32_t *>(p.sg.orig);
      aa.orig_bias = aa.orig_add = p.n_tri;
    }
    if (__builtin_amdgcn_ballot_w64(tb > 0.f && !in_box) != 0) return false;
    const f3 Pl = ld3(p.light_points + 4 * p.ll.point[li]);
    int n_tests = 0, sw = 0;
    if (!anyhit_sph_light_lists(p.ll, light_list_cell(p.ll, li, Pl, so),
                                SmemFetch<PairG>{reinterpret_cast<const PairG *>(p.sg.sorted2)}, p.n_tri, so, sL,
                                aa, n_tests, sw))
      return false;
    sw_all += sw;
    if (tb > 0.f) n_open = (n_open + (n_tests >> 3)) | kListServed;
    if (aa.kocc >= 0) {
      out.tocc = aa.tocc;
      out.kocc = aa.kocc;
    }
  }
  a.kocc = out.kocc;
  a.tocc = out.tocc;
  n_swept += sw_all;
  if (last_light) { // (shadow_sweep_smem's epilogue)
    const bool grp = p.sg.n_grp > 0;
    if (grp || tgrp) grp_open = n_open;
    tri_groups = tgrp;
    sph_groups = grp;
  }
  return true;
}

template <int STAGE, bool TGRP = false>
__global__ void __launch_bounds__(256, STAGE == STAGE_LDS ? 4 : (TGRP ? 5 : 6)) k_shade(const RenderParams p) {
  typedef float V;
  constexpr int NV = 1;
  constexpr int TW = 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE == STAGE_LDS ? kLdsChunkBytes : 16];
  __shared__ int vote_word[4];
  int vote_k = 0;
  if (STAGE == STAGE_SMEM) wg_vote_init(vote_word);
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
  struct {
    float t, v;
    int32_t idx;
  } hr;
  hr.t = FLT_MAX;
  hr.v = 0.f;
  hr.idx = -1;
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
    if (inside) {
      const size_t px = (size_t)lr * p.W + w;
      hr.idx = p.hits.idx[px];
      if (hr.idx >= 0) {
        hr.t = p.hits.t[px];
        if (p.tri_n) hr.v = p.hits.v[px];
      }
    }
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
  // any-hit tests the reference would have executed: summed over the WAVE light by light (a
  // per-lane 64-bit accumulator across the sweeps cost 12 B of scratch per lane)
  unsigned long long n_any = 0;
  int n_swept = 0;              // primitives this WAVE swept in any-hit loops (x64 = lane-tests)
  for (int li = 0; li < p.n_lights; ++li) {
    const DevLight Lt = p.lights[li];
    Any a[1];
    int grp_open = 0; // 8-record openings of group sweeps this ray needed (counters only)
    bool tri_groups = false, sph_groups = false; // this light swept triangle / sphere groups
    uint32_t cnt_lane = 0;                       // this ray's any-hit tests for this light
    f3 ro = N, rL = N; // shadow-ray origin (main.cpp:757 `hit`) and unit direction
    f3 lP = N;         // the light sample point and its index (light bins, ESC_STAGE_BVH)
    int lpt = 0;
    a[0].tb = 0.f;
    a[0].tocc = 0.f;
    a[0].kocc = -1;
    a[0].orig = nullptr;
    a[0].orig_bias = a[0].orig_add = 0;
    if (has_hit) {
      int ww = w, hh = h;
      if constexpr (STAGE == STAGE_SMEM) { // not kept live across the sweeps (tile_again)
        const Tile<1> Ta = tile_again(p);
        ww = Ta.w0 + Ta.lx0;
        hh = Ta.h_tile + Ta.ly;
      }
      // x % 1 == 0: a one-face light needs no draw (wave-uniform shortcut)
      const uint32_t face =
          (p.face_mode == 0) ? (uint32_t)p.fixed_face
          : (Lt.n_faces == 1) ? 0u
                              : face_hash(p.seed, (uint32_t)(hh * p.W + ww), (uint32_t)li,
                                          (uint32_t)Lt.n_faces);
      const f3 P = ld3(p.light_points + 4 * (Lt.first_point + (int)face)); // quirk S2
      lP = P;
      lpt = Lt.first_point + (int)face;
      // the primary direction is recomputed here (same ops, same bits) rather than kept in
      // registers across the any-hit loops of the previous light
      const f3 dir = (STAGE == STAGE_BVH) ? dir_kept : primary_dir(p, ww, hh);
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
        cnt_lane = (uint32_t)n_tests;
      } else if constexpr (STAGE == STAGE_SMEM) {
        bool served = false;
        if (shadow_lists_cover<TGRP>(p, li)) { // launch-uniform (see k_frame)
          Any af = a[0];
          int go = 0, sw = 0;
          bool tg = false, sgp = false;
          const bool ok = shadow_wave_lists<TGRP>(p, li, ro, rL, af, go, tg, sgp, sw);
          if (!wg_vote_any(vote_word, vote_k, !ok)) {
            a[0].kocc = af.kocc;
            a[0].tocc = af.tocc;
            grp_open = go;
            tri_groups = tg;
            sph_groups = sgp;
            n_swept += sw;
            served = true;
          }
        }
        if (!served)
          shadow_sweep_smem<TGRP>(p, lds_rays, tid, wave, lane, li, ro, rL, a, N, r, g, b, t, mi, grp_open,
                                  tri_groups, sph_groups, n_swept);
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
        if (STAGE != STAGE_BVH) {
          // index-order sweeps: up to and including the occluder.  Group sweeps: the hyper-groups
          // up to the occluder's (or all), 8 more filter tests per opening this ray needed.
          const int k = a[0].kocc;
          const bool by_tri = k >= 0 && k < p.n_tri;
          unsigned cnt = (grp_open & kTriListServed)
                             ? 0u // (its light's triangle lists served it: the batches it was live for)
                             : tri_groups ? (unsigned)(by_tri ? k / (kTriGroup * kTriSuper * kTriHyper) + 1
                                                              : p.tg.n_hyp)
                                          : (unsigned)(by_tri ? k + 1 : p.n_tri);
          // (a ray its light's lists served: the 8-sphere batches it was live for, nothing else)
          if (!by_tri && !(grp_open & kListServed))
            cnt += sph_groups ? (unsigned)(k >= 0 ? ((k - p.n_tri) >> 9) + 1 : p.sg.n_hyp)
                              : (unsigned)(k >= 0 ? k - p.n_tri + 1 : p.n_sph);
          cnt_lane = cnt + 8u * (unsigned)(grp_open & ~(kListServed | kTriListServed));
        }
      }
      if (p.shadows && a[0].kocc >= 0) {
        t = a[0].tocc; // occlusion() wrote the occluder's t2 through its reference (quirk S3)
      } else {         // :772-773 `continue` otherwise
        phong_add(p.mat[mi], N, rL, nl, r, g, b); // :768-788
      }
    }
    if constexpr (STAGE == STAGE_SMEM) {
      if (p.counters && p.shadows) n_any += wave_sum(cnt_lane); // wave-uniform
    } else {
      n_any += cnt_lane; // per lane (these variants have the registers)
    }
  }

  if constexpr (STAGE == STAGE_SMEM) { // tile fields recomputed, not kept live (tile_again)
    const Tile<1> Te = tile_again(p);
    const int tid_e = Te.wave * 64 + Te.lane;
    const bool inside_e = (Te.lr0 + Te.ly < p.n_local_rows) && (Te.h_tile + Te.ly < p.H) &&
                          (Te.w0 + Te.lx0 < p.W);
    emit_counters(p, tid_e, Te.lane, inside_e, has_hit, n_shadow, Te.lane == 0 ? n_any : 0ull,
                  (unsigned long long)n_swept * 64ull, true);
    write_tile(p, Te, tid_e, r, g, b, inside_e, lds_px);
  } else {
    emit_counters(p, tid, lane, inside, has_hit, n_shadow, n_any, (unsigned long long)n_swept * 64ull,
                  true);
    write_tile(p, T, tid, r, g, b, inside, lds_px);
  }
}

// ---------------------------------------------------------------------------------------
// k_frame: the whole frame of the default path in ONE kernel -- k_primary's closest hit (2 pixels per
// lane: a workgroup owns a 64 x 8 tile) followed, for each of the lane's two pixels in turn, by
// k_shade<SMEM>'s shading with the workgroup's 256 rays re-packed between segments.  What it saves
// over the two kernels: the hit planes (4-12 B per pixel written and read back: 2.2x the
// algorithmic HBM traffic on c4 became 1.1x), the primary direction computed three times per hit
// pixel (two divides, a square root and three more divides each), one launch.  The two halves
// still want different register budgets; they meet through LDS: after the sweep every lane parks
// dir, t, v, idx of both pixels (12 KB per workgroup) and the shading loop -- not unrolled, so one
// copy of its code -- picks one pixel up at a time and stores its finished colour straight from
// registers: the 16 lanes of a tile row write 192 consecutive, 64-byte aligned bytes of fp32 (48 of
// PPM bytes).  (Collecting the tile in LDS for whole-row store instructions cost 0.03 ms of a c4
// frame in index arithmetic, LDS round trips and a barrier.)  Same functions as the two-kernel path
// (primary_closest, shadow_sweep_smem, phong_add): same arithmetic, same image.
// ESC_RENDER_TWO_KERNELS keeps k_primary + k_shade.
// ---------------------------------------------------------------------------------------
DEVINL Tile<2> tile_again2(const RenderParams &p) { // (see tile_again); k_frame's (tiles_x, tiles_y) grid
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  Tile<2> T(p, (int)blockIdx.x, (int)blockIdx.y, t);
  T.wave = __builtin_amdgcn_readfirstlane(T.wave);
  return T;
}

// counters of a workgroup whose lanes carry several pixels: per-lane counts in, one atomic per
// counter and workgroup out (emit_counters' scheme).  All 256 threads (barriers inside).
DEVINL void emit_counters_n(const RenderParams &p, int tid, int lane, uint32_t n_inside, uint32_t n_hit,
                            uint32_t n_shadow, unsigned long long n_any_wave,
                            unsigned long long lane_tests_wave) {
  if (!p.counters) return;
  __shared__ unsigned long long wg_cnt[5];
  if (tid < 5) wg_cnt[tid] = 0ull;
  __syncthreads();
  uint32_t ni = n_inside, nh = n_hit, ns = n_shadow;
  for (int o = 32; o > 0; o >>= 1) {
    ni += __shfl_down(ni, o);
    nh += __shfl_down(nh, o);
    ns += __shfl_down(ns, o);
  }
  if (lane == 0) {
    atomicAdd(&wg_cnt[0], (unsigned long long)ni);
    atomicAdd(&wg_cnt[1], (unsigned long long)nh);
    atomicAdd(&wg_cnt[2], (unsigned long long)ns);
    atomicAdd(&wg_cnt[3], n_any_wave);
    atomicAdd(&wg_cnt[4], lane_tests_wave);
  }
  __syncthreads();
  if (tid < 5 && wg_cnt[tid])
    atomicAdd(&p.counters[(blockIdx.x % kCounterSets) * 8 + tid], wg_cnt[tid]);
}

DEVINL uint8_t quantise_channel(float c) { // main.cpp:676-682 clamp > 1, int(c * 255)
  const float cc = (c > 1.f) ? 1.f : c;
  return (uint8_t)(int)(cc * 255.f);
}

template <bool GRP, bool TGRP>
__global__ void __launch_bounds__(256, TGRP ? 4 : 5) k_frame(const RenderParams p) {
  __shared__ RepackLds lds_rays;
  __shared__ float park[2][6][256]; // per pixel q of thread t: dir xyz, t, v, idx
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[16];
  __shared__ int vote_word[4];
  int vote_k = 0;
  wg_vote_init(vote_word);
  const int tid = threadIdx.x;
  { // ---- camera.h:31-34 + main.cpp:722: both pixels of every lane (k_primary's body)
    const Tile<2> T(p, Tile<2>::Grid2D{});
    const int h = T.h_tile + T.ly;
    f3 dir[2];
    Hit hit[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      dir[q] = primary_dir(p, T.w0 + T.lx0 + 16 * q, h);
      hit[q].t = FLT_MAX; // main.cpp:715
      hit[q].v = 0.f;
      hit[q].idx = -1;
    }
    V3<v2f> dv[1];
    pack3<v2f, 1>(dir, dv);
    primary_closest<STAGE_SMEM, v2f, 1, GRP, 2>(p, T, dv, hit, lds_raw);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      park[q][0][tid] = dir[q].x;
      park[q][1][tid] = dir[q].y;
      park[q][2][tid] = dir[q].z;
      park[q][3][tid] = hit[q].t;
      park[q][4][tid] = hit[q].v;
      park[q][5][tid] = __int_as_float(hit[q].idx);
    }
  }
  // (every slot above is read back by the thread that wrote it: no barrier needed until the store)
  const f3 origin = mk(p.origin[0], p.origin[1], p.origin[2]);
  const float nl = (float)p.n_lights;
  uint32_t n_inside = 0, n_hit = 0, n_shadow = 0;
  unsigned long long n_any = 0; // wave-uniform
  int n_swept = 0;
#pragma unroll 1
  for (int q = 0; q < 2; ++q) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    bool has_hit;
    float t;
    f3 N = mk(0.f, 0.f, 0.f);
    int mi = 0;
    { // ---- main.cpp:723-738 normal of the hit
      const Tile<2> Ta = tile_again2(p);
      const bool inside = (Ta.lr0 + Ta.ly < p.n_local_rows) && (Ta.h_tile + Ta.ly < p.H) &&
                          (Ta.w0 + Ta.lx0 + 16 * q < p.W);
      const int idx = __float_as_int(park[q][5][tid]);
      has_hit = inside && idx >= 0;
      t = park[q][3][tid];
      n_inside += inside ? 1u : 0u;
      n_hit += has_hit ? 1u : 0u;
      if (has_hit) {
        if (idx < p.n_tri) {
          const DevTriFace Tf = p.tri_face[idx]; // :728-731, hoisted (k_prepare_face_normals)
          N = mk(Tf.n[0], Tf.n[1], Tf.n[2]);
          mi = Tf.geom;
          if (p.mat[mi].has_normals) { // :733-738 with u == 0 (quirk S1)
            const DevTriN Q = p.tri_n[idx];
            const float u = 0.f, v = park[q][4][tid];
            N = normalize((ld3(Q.n1) * u + ld3(Q.n2) * v) + ld3(Q.n0) * ((1.f - u) - v));
          }
        } else {
          const int k = idx - p.n_tri;
          const DevSph S = p.sph[k];
          const f3 dir = mk(park[q][0][tid], park[q][1][tid], park[q][2][tid]);
          N = normalize((origin + dir * t) - mk(S.cx, S.cy, S.cz)); // extension
          mi = p.sph_mat[k];
        }
      }
    }
    // ---- main.cpp:740-789 per-light shading (k_shade<SMEM>'s loop)
    float r = 0.f, g = 0.f, b = 0.f; // vec3 default ctor, main.cpp:557-558
    for (int li = 0; li < p.n_lights; ++li) {
      const DevLight Lt = p.lights[li];
      Any a[1];
      int grp_open = 0;
      bool tri_groups = false, sph_groups = false;
      uint32_t cnt_lane = 0;
      f3 ro = N, rL = N;
      a[0].tb = 0.f;
      a[0].tocc = 0.f;
      a[0].kocc = -1;
      a[0].orig = nullptr;
      a[0].orig_bias = a[0].orig_add = 0;
      if (has_hit) {
        uint32_t face = (p.face_mode == 0) ? (uint32_t)p.fixed_face : 0u;
        if (p.face_mode != 0 && Lt.n_faces != 1) { // x % 1 == 0: a one-face light needs no draw
          const Tile<2> Ta = tile_again2(p);
          face = face_hash(p.seed, (uint32_t)((Ta.h_tile + Ta.ly) * p.W + Ta.w0 + Ta.lx0 + 16 * q),
                           (uint32_t)li, (uint32_t)Lt.n_faces);
        }
        const f3 P = ld3(p.light_points + 4 * (Lt.first_point + (int)face)); // quirk S2
        const f3 dir = mk(park[q][0][tid], park[q][1][tid], park[q][2][tid]);
        ro = origin + dir * (t - FLT_EPSILON); // :757-758
        rL = P - ro;                           // :759
        const float len = length(rL);          // :761
        t = len - FLT_EPSILON;                 // :764
        rL = normalize(rL);                    // :766
        a[0].tb = t;
      }
      if (!(a[0].tb > 0.f)) a[0].tb = 0.f; // dead rays carry tb = 0
      if (p.shadows) {
        bool served = false;
        if (shadow_lists_cover<TGRP>(p, li)) { // launch-uniform
          Any af = a[0];
          int go = 0, sw = 0;
          bool tg = false, sgp = false;
          const bool ok = shadow_wave_lists<TGRP>(p, li, ro, rL, af, go, tg, sgp, sw);
          if (!wg_vote_any(vote_word, vote_k, !ok)) { // every wave of the workgroup was served
            a[0].kocc = af.kocc;
            a[0].tocc = af.tocc;
            grp_open = go;
            tri_groups = tg;
            sph_groups = sgp;
            n_swept += sw;
            served = true;
          }
        }
        if (!served)
          shadow_sweep_smem<TGRP>(p, lds_rays, tid, wave, lane, li, ro, rL, a, N, r, g, b, t, mi, grp_open,
                                  tri_groups, sph_groups, n_swept);
      }
      if (has_hit) {
        if (p.shadows && p.counters) { // (instrumentation: ESC_RENDER_NO_COUNTERS skips it)
          n_shadow += 1u;
          const int k = a[0].kocc; // (the counter rules of k_shade)
          const bool by_tri = k >= 0 && k < p.n_tri;
          unsigned cnt = (grp_open & kTriListServed)
                             ? 0u // (its light's triangle lists served it: the batches it was live for)
                             : tri_groups ? (unsigned)(by_tri ? k / (kTriGroup * kTriSuper * kTriHyper) + 1
                                                              : p.tg.n_hyp)
                                          : (unsigned)(by_tri ? k + 1 : p.n_tri);
          if (!by_tri && !(grp_open & kListServed))
            cnt += sph_groups ? (unsigned)(k >= 0 ? ((k - p.n_tri) >> 9) + 1 : p.sg.n_hyp)
                              : (unsigned)(k >= 0 ? k - p.n_tri + 1 : p.n_sph);
          cnt_lane = cnt + 8u * (unsigned)(grp_open & ~(kListServed | kTriListServed));
        }
        if (p.shadows && a[0].kocc >= 0) {
          t = a[0].tocc; // occlusion() wrote the occluder's t2 through its reference (quirk S3)
        } else {         // :772-773 `continue` otherwise
          phong_add(p.mat[mi], N, rL, nl, r, g, b); // :768-788
        }
      }
      if (p.counters && p.shadows) n_any += wave_sum(cnt_lane); // wave-uniform
    }
    // the fp32 pixel straight from its registers: the 16 lanes of a tile row write 192 consecutive
    // bytes, 64-byte aligned whenever W is a multiple of 16 (whole sectors; the other half of each
    // 128-byte line follows with the lane's second pixel)
    if (p.out_f32) {
      const Tile<2> Tw = tile_again2(p);
      const int lr = Tw.lr0 + Tw.ly, w = Tw.w0 + Tw.lx0 + 16 * q;
      if (lr < p.n_local_rows && Tw.h_tile + Tw.ly < p.H && w < p.W) {
        float *o = p.out_f32 + ((size_t)lr * p.W + w) * 3;
        o[0] = r;
        o[1] = g;
        o[2] = b;
      }
    }
    if (p.out_u8) { // main.cpp:676-682; three bytes per pixel, 48 consecutive bytes per tile row and pixel q
      const Tile<2> Tw = tile_again2(p);
      const int lr = Tw.lr0 + Tw.ly, w = Tw.w0 + Tw.lx0 + 16 * q;
      if (lr < p.n_local_rows && Tw.h_tile + Tw.ly < p.H && w < p.W) {
        uint8_t *o = p.out_u8 + ((size_t)lr * p.W + w) * 3;
        o[0] = quantise_channel(r);
        o[1] = quantise_channel(g);
        o[2] = quantise_channel(b);
      }
    }
  }
  {
    const Tile<2> Te = tile_again2(p);
    const int tid_e = Te.wave * 64 + Te.lane;
    emit_counters_n(p, tid_e, Te.lane, n_inside, n_hit, n_shadow, Te.lane == 0 ? n_any : 0ull,
                    Te.lane == 0 ? (unsigned long long)n_swept * 64ull : 0ull);

  }
}

// ---------------------------------------------------------------------------------------
// Queue form of the shadow pass (rt_device.h ShadeQueue): per light
//     k_shadow_setup  ->  k_anyhit_segment x (segments of the primitive list)  ->  k_shade_finish
// The arithmetic of every ray is the fused k_shade's, statement for statement; only WHERE a ray is
// tested (which wave, next to which other rays) differs.
// ---------------------------------------------------------------------------------------

// main.cpp:740-766 for light `li`: light sample and shadow ray of every hit pixel (the normal,
// main.cpp:723-738, is computed by k_shade_finish for the pixels that turn out unoccluded).
template <bool MULTI>
__global__ void __launch_bounds__(256) k_shadow_setup(const RenderParams p, int li) {
  const Tile<1> T(p);
  const int lr = T.lr0 + T.ly, h = T.h_tile + T.ly, w = T.w0 + T.lx0;
  const bool inside = (lr < p.n_local_rows) && (h < p.H) && (w < p.W);
  if (!inside) return;
  const size_t npx = (size_t)p.n_local_rows * p.W;
  const size_t px = (size_t)lr * p.W + w;
  const int32_t idx = p.hits.idx[px];
  if (idx < 0) return; // no record for a miss: the first segment reads the idx plane itself
  const f3 origin = mk(p.origin[0], p.origin[1], p.origin[2]);
  float t = (MULTI && li > 0) ? p.sq.state[px] : p.hits.t[px];
  const DevLight Lt = p.lights[li];
  const uint32_t face = (p.face_mode == 0) ? (uint32_t)p.fixed_face
                        : (Lt.n_faces == 1) ? 0u
                                            : face_hash(p.seed, (uint32_t)(h * p.W + w), (uint32_t)li,
                                                        (uint32_t)Lt.n_faces);
  const f3 P = ld3(p.light_points + 4 * (Lt.first_point + (int)face)); // quirk S2
  const f3 ro = origin + primary_dir(p, w, h) * (t - FLT_EPSILON);     // :757-758
  f3 rL = P - ro;                                                        // :759
  const float len = length(rL);                                          // :761
  t = len - FLT_EPSILON;                                                 // :764
  rL = normalize(rL);                                                    // :766
  ShadowRay R;
  R.ox = ro.x;
  R.oy = ro.y;
  R.oz = ro.z;
  R.tb = (t > 0.f) ? t : 0.f; // dead rays carry tb = 0
  R.lx = rL.x;
  R.ly = rL.y;
  R.lz = rL.z;
  R.kocc = -1;
  p.sq.rays[px] = R;
  if (MULTI) p.sq.state[px] = t; // what main.cpp:764 leaves in `t` for the next light
  (void)npx;
}

// One segment of occlusion()'s loop (main.cpp:314-329) for every ray still looking.
// Workgroups pull batches of 64-ray chunks from the input queue (one atomic per batch), each wave
// sweeps the segment for its chunks, decided rays write their occluder back to their ShadowRay,
// undecided ones are staged in LDS and leave as full chunks of the output queue (one atomic per
// flush).  Every workgroup leaves the loop when the cursor passes the end: nothing waits on
// another workgroup.
constexpr int kSegBatch = 2; // chunks per wave per fetch (1: 10.59, 2: 10.23, 4: 10.36 ms per c4 frame)
template <bool HAS_TRI, bool HAS_SPH>
__global__ void __launch_bounds__(256) k_anyhit_segment(const RenderParams p, const SegArgs a) {
  __shared__ uint32_t stage[64 + 4 * kSegBatch * 64];
  __shared__ uint32_t n_staged, batch0, out_base;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const uint32_t n_chunks = a.qin ? *a.in_chunks : a.n_identity_chunks;
  if (tid == 0) n_staged = 0;
  int swept = 0; // primitives this wave swept (x64 = lane-tests)
  for (;;) {
    __syncthreads(); // n_staged / stage of the previous round; batch0 readers of the previous round
    // (fetching the next batch early, while this one is worked on, was measured: 4.70 instead of
    // 4.44 ms -- wave 0's ray loads queue up behind the atomic's return)
    if (tid == 0) batch0 = atomicAdd(a.cursor, 4u * kSegBatch);
    __syncthreads();
    const uint32_t c0 = batch0;
    if (c0 >= n_chunks) break; // workgroup-uniform
#pragma unroll 1
    for (int j = 0; j < kSegBatch; ++j) {
      const uint32_t c = c0 + (uint32_t)(wave * kSegBatch + j);
      if (c >= n_chunks) break; // wave-uniform
      uint32_t id;
      bool valid;
      if (a.qin) {
        id = a.qin[(size_t)c * 64 + lane];
        valid = id != kQueueInvalid;
      } else {
        id = c * 64u + (uint32_t)lane;
        valid = id < a.n_pixels && p.hits.idx[id] >= 0;
      }
      Any aa[1];
      aa[0].tb = 0.f;
      aa[0].tocc = 0.f;
      aa[0].kocc = -1;
      aa[0].orig = nullptr;
      aa[0].orig_bias = aa[0].orig_add = 0;
      f3 so = mk(0.f, 0.f, 0.f), sL = so;
      if (valid) {
        const ShadowRay R = p.sq.rays[id];
        so = mk(R.ox, R.oy, R.oz);
        sL = mk(R.lx, R.ly, R.lz);
        aa[0].tb = R.tb;
      }
      if (__builtin_amdgcn_ballot_w64(aa[0].tb > 0.f)) {
        if (HAS_TRI) {
          const V3<float> sov[1] = {{so.x, so.y, so.z}}, sLv[1] = {{sL.x, sL.y, sL.z}};
          swept += a.tri_count; // upper bound: exits inside a segment are not subtracted
          if (p.use_filter && a.tri_count >= 8) { // tri_first is even (host: segments are)
            bool far;
            const RayTF rt = make_ray_tri_filter(so, sL, p.shadow_center, p.shadow_rho_max, far);
            const RayF rs = make_ray_filter(so, sL, p.shadow_center);
            anyhit_tri_filter(SmemFetch<TriPairPF>{reinterpret_cast<const TriPairPF *>(p.tri2_pf) +
                                                   (a.tri_first >> 1)},
                              SmemFetch<TriPairF>{reinterpret_cast<const TriPairF *>(p.tri2_f) +
                                                  (a.tri_first >> 1)},
                              SmemFetch<DevTri>{p.tri + a.tri_first}, a.tri_count, a.tri_first, so,
                              sL, rs, rt, far, aa);
          } else {
            anyhit_tri<float, 1>(SmemFetch<DevTri>{p.tri + a.tri_first}, a.tri_count, a.tri_first,
                                 sov, sLv, aa);
          }
        }
        if (HAS_SPH) {
          if (p.use_filter) {
            const RayF rf = make_ray_filter(so, sL, p.shadow_center);
            swept += 2 * anyhit_sph_pairs_filter(
                             SmemFetch<PairF>{reinterpret_cast<const PairF *>(a.sph2_f) + a.rec_first},
                             SmemFetch<PairG>{reinterpret_cast<const PairG *>(a.sph2) + a.rec_first},
                             a.rec_count, p.n_tri + 2 * a.rec_first, so, sL, rf, aa[0]);
          } else {
            swept += 2 * anyhit_sph_pairs(
                             SmemFetch<PairG>{reinterpret_cast<const PairG *>(a.sph2) + a.rec_first},
                             a.rec_count, p.n_tri + 2 * a.rec_first, so, sL, aa[0]);
          }
        }
      }
      if (aa[0].kocc >= 0) { // decided: occlusion() returned true and wrote t2 through its reference
        ShadowRay *R = p.sq.rays + id;
        R->ox = aa[0].tocc;
        R->tb = 0.f;
        R->kocc = aa[0].kocc;
      }
      if (a.qout) {
        const bool survive = aa[0].tb > 0.f;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(survive);
        if (m) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&n_staged, (uint32_t)__popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          if (survive) stage[base + __popcll(m & ((1ull << lane) - 1ull))] = id;
        }
      }
    }
    if (a.qout) { // full chunks leave for the output queue, the remainder moves to the front
      __syncthreads();
      const uint32_t ns = n_staged, full = ns >> 6, rem = ns & 63u;
      if (full) {
        if (tid == 0) out_base = atomicAdd(a.out_chunks, full);
        __syncthreads();
        for (uint32_t i = tid; i < full * 64u; i += 256u) a.qout[(size_t)out_base * 64 + i] = stage[i];
        const uint32_t keep = (tid < (int)rem) ? stage[full * 64u + tid] : 0u;
        __syncthreads();
        if (tid < (int)rem) stage[tid] = keep;
        if (tid == 0) n_staged = rem;
      }
    }
  }
  if (a.qout) { // the break above is workgroup-uniform and follows a barrier: n_staged is settled
    const uint32_t ns = n_staged;
    if (ns) {
      if (tid == 0) out_base = atomicAdd(a.out_chunks, 1u);
      __syncthreads();
      if (tid < 64) a.qout[(size_t)out_base * 64 + tid] = (tid < (int)ns) ? stage[tid] : kQueueInvalid;
    }
  }
  emit_counters(p, tid, lane, false, false, 0u, 0ull, (unsigned long long)swept * 64ull, false);
}

// main.cpp:768-789 for light `li` (+ the framebuffer after the last light)
template <bool MULTI>
__global__ void __launch_bounds__(256) k_shade_finish(const RenderParams p, int li, int last) {
  __shared__ float lds_px[32 * kTileH * 3];
  const Tile<1> T(p);
  const int tid = threadIdx.x, lane = T.lane;
  const int lr = T.lr0 + T.ly, h = T.h_tile + T.ly, w = T.w0 + T.lx0;
  const bool inside = (lr < p.n_local_rows) && (h < p.H) && (w < p.W);
  const size_t npx = (size_t)p.n_local_rows * p.W;
  const size_t px = inside ? (size_t)lr * p.W + w : 0;
  const bool has_hit = inside && p.hits.idx[px] >= 0;
  float r = 0.f, g = 0.f, b = 0.f; // vec3 default ctor, main.cpp:557-558
  uint32_t n_shadow = 0;
  unsigned long long n_any = 0;
  if (has_hit) {
    if (MULTI && li > 0) {
      r = p.sq.state[npx + px];
      g = p.sq.state[2 * npx + px];
      b = p.sq.state[3 * npx + px];
    }
    const ShadowRay R = p.sq.rays[px];
    n_shadow = 1u;
    n_any = (R.kocc >= 0) ? (unsigned)(R.kocc + 1) : (unsigned)(p.n_tri + p.n_sph);
    if (R.kocc >= 0) {
      if (MULTI) p.sq.state[px] = R.ox; // occlusion() wrote the occluder's t2 into t (quirk S3)
    } else {                            // :772-773 `continue` otherwise
      // main.cpp:723-738 normal of the hit (per-lane gathers; only unoccluded pixels get here)
      const int32_t idx = p.hits.idx[px];
      f3 N;
      int mi;
      if (idx < p.n_tri) {
        const DevTri Tr = p.tri[idx];
        N = normalize(cross(ld3(Tr.e1), ld3(Tr.e2))); // :728-731
        mi = Tr.geom;
        if (p.mat[mi].has_normals) { // :733-738 with u == 0 (quirk S1)
          const DevTriN Q = p.tri_n[idx];
          const float u = 0.f, v = p.hits.v[px];
          N = normalize((ld3(Q.n1) * u + ld3(Q.n2) * v) + ld3(Q.n0) * ((1.f - u) - v));
        }
      } else {
        const int k = idx - p.n_tri;
        const DevSph S = p.sph[k];
        const f3 origin = mk(p.origin[0], p.origin[1], p.origin[2]);
        N = normalize((origin + primary_dir(p, w, h) * p.hits.t[px]) - mk(S.cx, S.cy, S.cz));
        mi = p.sph_mat[k];
      }
      const f3 rL = mk(R.lx, R.ly, R.lz);
      const float nl = (float)p.n_lights;
      phong_add(p.mat[mi], N, rL, nl, r, g, b); // :768-788
    }
    if (MULTI && !last) {
      p.sq.state[npx + px] = r;
      p.sq.state[2 * npx + px] = g;
      p.sq.state[3 * npx + px] = b;
    }
  }
  emit_counters(p, tid, lane, inside, has_hit, n_shadow, n_any, 0ull, last != 0);
  if (last) write_tile(p, T, tid, r, g, b, inside, lds_px);
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
namespace esc {
__global__ void __launch_bounds__(256) k_prepare_face_normals(const DevTri *tri, DevTriFace *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const DevTri Tr = tri[i];
  const f3 N = normalize(cross(ld3(Tr.e1), ld3(Tr.e2))); // main.cpp:728-731, the shading kernels' own line
  DevTriFace F;
  F.n[0] = N.x;
  F.n[1] = N.y;
  F.n[2] = N.z;
  F.geom = Tr.geom;
  out[i] = F;
}
} // namespace esc
extern "C" int esc_launch_face_normals(const esc::DevTri *tri, esc::DevTriFace *out, int n, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(esc::k_prepare_face_normals, dim3((n + 255) / 256), dim3(256), 0, stream, tri, out, n);
  return (int)hipGetLastError();
}

extern "C" int esc_launch_prepare(const esc::RenderParams *p, esc::DevTriP *tri_p,
                                  esc::DevTriF *tri_f, esc::DevTriPF *tri_pf, esc::DevSphP *sph_p,
                                  esc::DevSphF *sph_f, const esc::SphGroups *sg,
                                  const esc::TriGroups *tg, hipStream_t stream) {
  const int n = p->n_tri > p->n_sph ? p->n_tri : p->n_sph;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(esc::k_prepare_primary, dim3((n + 255) / 256), dim3(256), 0, stream, p->tri,
                     tri_p, tri_f, tri_pf, p->n_tri, p->sph, sph_p, sph_f, p->n_sph, p->origin[0],
                     p->origin[1], p->origin[2]);
  if (tg->n_grp > 0) {
    hipLaunchKernelGGL(esc::k_prepare_tri_groups, dim3((tg->n_grp * esc::kTriGroup + 255) / 256),
                       dim3(256), 0, stream, *tg, p->origin[0], p->origin[1], p->origin[2]);
    hipLaunchKernelGGL(esc::k_prepare_tri_merge, dim3((2 * tg->n_sup + 255) / 256), dim3(256), 0, stream,
                       *tg, 1, p->origin[0], p->origin[1], p->origin[2]);
    hipLaunchKernelGGL(esc::k_prepare_tri_merge, dim3((tg->n_hyp + 255) / 256), dim3(256), 0, stream,
                       *tg, 2, p->origin[0], p->origin[1], p->origin[2]);
  }
  if (sg->n_grp > 0)
    hipLaunchKernelGGL(esc::k_prepare_groups, dim3((sg->n_grp * esc::kSphGroup + 255) / 256),
                       dim3(256), 0, stream, *sg, p->origin[0], p->origin[1], p->origin[2]);
  return (int)hipGetLastError();
}

// tile lists of the primary pass (rt_lists.h): hdr / cnt zeroed by the caller on this stream; after
// esc_launch_prepare (the triangle groups' frame cones)
extern "C" int esc_launch_tile_lists(const esc::RenderParams *p, hipStream_t stream) {
  if (p->sl.enabled && p->sg.n_grp > 0)
    hipLaunchKernelGGL(esc::k_bin_spheres, dim3(p->sg.n_grp * esc::kSphGroup / 4), dim3(256), 0, stream, *p);
  if (p->tl.enabled && p->tg.n_grp > 0) {
    hipLaunchKernelGGL(esc::k_bin_triangles, dim3(p->tg.n_grp * esc::kTriGroup / 4), dim3(256), 0, stream, *p);
    hipLaunchKernelGGL(esc::k_bin_tri_escape, dim3((unsigned)p->tl.tile_rows), dim3(256), 0, stream, *p);
  }
  return (int)hipGetLastError();
}

// light lists of the shadow pass (rt_lists.h): hdr / cnt zeroed by the caller on this stream
extern "C" int esc_launch_light_lists(const esc::RenderParams *p, hipStream_t stream) {
  const int n_rec = p->sg.n_grp * (esc::kSphGroup / 2);
  if (p->ll.enabled && p->ll.n_listed > 0 && n_rec > 0) {
    hipLaunchKernelGGL(esc::k_bin_light_pairs, dim3((n_rec + 3) / 4), dim3(256), 0, stream, *p);
    const int n_cells = p->ll.n_listed * 6 * p->ll.R * p->ll.R;
    hipLaunchKernelGGL(esc::k_sort_light_cells<false>, dim3((n_cells + 255) / 256), dim3(256), 0, stream, *p);
  }
  const int n_trec = p->tg.n_grp * (esc::kTriGroup / 2);
  if (p->lt.enabled && p->lt.n_listed > 0 && n_trec > 0) {
    hipLaunchKernelGGL(esc::k_bin_light_tri_pairs, dim3((n_trec + 3) / 4), dim3(256), 0, stream, *p);
    const int n_cells = p->lt.n_listed * 6 * p->lt.R * p->lt.R;
    hipLaunchKernelGGL(esc::k_bin_light_tri_escape, dim3((n_cells + 255) / 256), dim3(256), 0, stream, *p);
    hipLaunchKernelGGL(esc::k_sort_light_cells<true>, dim3((n_cells + 255) / 256), dim3(256), 0, stream, *p);
  }
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
  if (STAGE == esc::STAGE_SMEM && NV * esc::lanes_of<V>::n == 2 && p->use_filter &&
      (p->sg.n_grp > 0 || p->tg.n_grp > 0))
    hipLaunchKernelGGL((esc::k_primary<STAGE, V, NV, true>), dim3(tiles_x * tiles_y), dim3(256), 0, stream,
                       *p);
  else
    hipLaunchKernelGGL((esc::k_primary<STAGE, V, NV>), dim3(tiles_x * tiles_y), dim3(256), 0, stream, *p);
}

// the queue form of the shadow pass for one light: setup, one launch per segment, finish.
// segs: (tri_first, tri_count, rec_first, rec_count) per segment; ctl: this light's control words, 2 per segment, zeroed.
extern "C" int esc_launch_shade_queue(const esc::RenderParams *p, int li, int last, const int *segs,
                                      int n_segs, uint32_t *ctl, int n_wg, hipStream_t stream) {
  const int tiles_y = (p->n_local_rows + esc::kTileH - 1) / esc::kTileH;
  const int grid = ((p->W + 31) / 32) * tiles_y;
  const bool multi = p->n_lights > 1;
  if (multi)
    hipLaunchKernelGGL((esc::k_shadow_setup<true>), dim3(grid), dim3(256), 0, stream, *p, li);
  else
    hipLaunchKernelGGL((esc::k_shadow_setup<false>), dim3(grid), dim3(256), 0, stream, *p, li);
  const uint32_t n_pixels = (uint32_t)((size_t)p->n_local_rows * p->W);
  for (int s = 0; s < n_segs; ++s) {
    esc::SegArgs a;
    a.qin = (s == 0) ? nullptr : p->sq.q[(s - 1) & 1];
    a.in_chunks = (s == 0) ? nullptr : ctl + 2 * (s - 1);
    a.n_identity_chunks = (n_pixels + 63u) / 64u;
    a.n_pixels = n_pixels;
    a.qout = (s == n_segs - 1) ? nullptr : p->sq.q[s & 1];
    a.out_chunks = ctl + 2 * s;
    a.cursor = ctl + 2 * s + 1;
    a.tri_first = segs[4 * s + 0];
    a.tri_count = segs[4 * s + 1];
    a.rec_first = segs[4 * s + 2];
    a.rec_count = segs[4 * s + 3];
    const bool ord = last && p->sph2_ord != nullptr;
    a.sph2 = ord ? p->sph2_ord : p->sph2;
    a.sph2_f = ord ? p->sph2_f_ord : p->sph2_f;
    if (a.tri_count && a.rec_count)
      hipLaunchKernelGGL((esc::k_anyhit_segment<true, true>), dim3(n_wg), dim3(256), 0, stream, *p, a);
    else if (a.tri_count)
      hipLaunchKernelGGL((esc::k_anyhit_segment<true, false>), dim3(n_wg), dim3(256), 0, stream, *p, a);
    else
      hipLaunchKernelGGL((esc::k_anyhit_segment<false, true>), dim3(n_wg), dim3(256), 0, stream, *p, a);
  }
  if (multi)
    hipLaunchKernelGGL((esc::k_shade_finish<true>), dim3(grid), dim3(256), 0, stream, *p, li, last);
  else
    hipLaunchKernelGGL((esc::k_shade_finish<false>), dim3(grid), dim3(256), 0, stream, *p, li, last);
  return (int)hipGetLastError();
}

// the primary pass alone (the queue form launches its own shading kernels)
extern "C" int esc_launch_primary_only(const esc::RenderParams *p, int px, hipStream_t stream) {
  if (p->n_local_rows <= 0 || p->W <= 0) return 0;
  using esc::v2f;
  // SMEM staging always carries 2 pixels per lane: that is the variant the packed filter bodies are
  // written for; the 1- and 4-pixel variants spilled SGPRs / VGPRs to scratch and
  // were removed.  The result does not depend on it.
  (void)px;
  launch_primary<esc::STAGE_SMEM, v2f, 1>(p, stream);
  return (int)hipGetLastError();
}

// stage: 1 SMEM (always 2 px), 2 LDS, 3 BVH (px ignored).  px: pixels per work-item of the LDS primary pass (1, 2 or 4); the shade
// pass always carries one pixel per work-item.
extern "C" int esc_launch_render(const esc::RenderParams *p, int stage, int px, hipStream_t stream,
                                 hipEvent_t between, int two_kernels) {
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
    if (between) (void)hipEventRecord(between, stream);
    hipLaunchKernelGGL((esc::k_shade<esc::STAGE_LDS>), dim3(shade_grid), dim3(256), 0, stream, *p);
  } else if (!two_kernels && !between) { // the whole frame in one kernel (k_frame)
    const dim3 grid((unsigned)((p->W + 63) / 64), (unsigned)tiles_y); // (x, y) = the tile: Tile<2>::Grid2D
    const bool grp = p->use_filter && (p->sg.n_grp > 0 || p->tg.n_grp > 0);
    const bool tgrp = p->use_filter && p->tg.n_grp > 0;
    if (tgrp)
      hipLaunchKernelGGL((esc::k_frame<true, true>), grid, dim3(256), 0, stream, *p);
    else if (grp)
      hipLaunchKernelGGL((esc::k_frame<true, false>), grid, dim3(256), 0, stream, *p);
    else
      hipLaunchKernelGGL((esc::k_frame<false, false>), grid, dim3(256), 0, stream, *p);
  } else {
    launch_primary<esc::STAGE_SMEM, v2f, 1>(p, stream); // always 2 px: see esc_launch_primary_only
    if (between) (void)hipEventRecord(between, stream);
    if (p->use_filter && p->tg.n_grp > 0)
      hipLaunchKernelGGL((esc::k_shade<esc::STAGE_SMEM, true>), dim3(shade_grid), dim3(256), 0, stream, *p);
    else
      hipLaunchKernelGGL((esc::k_shade<esc::STAGE_SMEM>), dim3(shade_grid), dim3(256), 0, stream, *p);
  }
  return (int)hipGetLastError();
}
