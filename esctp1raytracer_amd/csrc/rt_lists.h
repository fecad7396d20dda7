// rt_lists.h -- tile lists for primary rays (rt_device.h TileLists): the kernels that build them
// and the list-driven form of the primary sweep.  Included by rt_kernels.hip only.
//
// Tile lists.  A wave of the primary pass owns a tile of 32 x 4 pixels.  Instead of sweeping the
// hyper-groups and opening super-groups, groups and members on the way down, it reads the list of
// the PRIMITIVES (slots of the group-sorted tables) some ray of its tile may have to test and runs
// the member code of the sweep on them: spheres straight through the reference arithmetic
// (test_sph_primary_sorted), triangles through the triangle filter and, behind it, the reference
// arithmetic (tri2_primary_filter_pk, test_tri2_primary_sorted).  What makes a list complete is the
// statement the filters already rest on (rt_brute.h "Sphere GROUPS", "Triangle pre-filter",
// "Triangle GROUPS" (P)), for a ray from the camera o with the reference's fp32 direction d:
//   sphere i not rejected at `disc < 0`  ==>  the LINE o + l d passes within
//        D_i = r_i (1+u) + 4.41 sqrt(u) |oc_i|_1 + 2^-74   of c~_i = o - fl(o - c_i);
//   triangle t accepted  ==>  (S_t) the line through o'_t meets t's plane at a point X* within rho_t
//        of t IN THAT PLANE (X* = v0 + u* e1 + v* e2), o'_t within 1.01u |tvec|_1 of o,
//        or (E_t) |d . n_t| < tau_t / |n1_t| -- and (P) that only with the camera within H_t of the plane.
// All three are regions of the image plane:
//  * "The line along p passes within R of c = C - o" for p(s, t) = A + s H + t V (A = llc - o, s =
//    w / (W-1), t = h / (H-1): camera.h:31-34) is p^T M p <= 0 with M = (|c|^2 - R^2) I - c c^T, a
//    conic in (s, t).  With x = (s, t, 1), p = B x, B = [H V A], the lines s = const tangent to it
//    solve Qi_33 s^2 - 2 Qi_13 s + Qi_11 = 0 with Qi = (B^T M B)^-1 ~ B^-1 (I - c c^T / R^2) B^-T, i.e.
//    Qi_jk = b_j . b_k - (b_j . c)(b_k . c) / R^2 over the rows b_1, b_2, b_3 of B^-1 (t: b_2).  Qi_33 < 0
//    <=> the sphere lies wholly on one side of the camera plane parallel to the image; otherwise the
//    region is unbounded and the primitive goes to the short list every tile tests.  The double
//    cone has both nappes, so a sphere BEHIND the camera is binned where its lines cross the image
//    -- more than needed (an accept needs t2 > 0), never less.  Evaluated in double from the fp32
//    inputs; what fp32 rounding does to the rays grows the rectangle: the reference evaluates
//    ((llc + H s) + V t) - o in fp32 and normalises, which moves p by at most eps_p = 2^-19 (|llc|_1 +
//    |H|_1 + |V|_1 + |o|_1) (16u would do), i.e. the ray's (s, t) by at most (|b_1| + |b_3|) eps_p 1.01
//    resp. (|b_2| + |b_3|) eps_p 1.01; one more pixel covers fl(w / (W-1)) and the floor / ceil
//    (rt_tile_math.h; tests/test_tile_lists.py checks it per pixel by brute force).
//  * (S_t): the in-plane disc of radius rho' around a point lies in the in-plane square of half-side
//    rho', so t dilated by rho' in its plane lies in the convex hull of the 12 corners v_i +- rho' e_a
//    +- rho' e_b (e_a, e_b orthonormal in the plane), and the rays that meet a convex hull are those
//    through the hull of its projection (all 12 on one side of the camera plane; else the triangle
//    is global): the rectangle is the union of the 12 points' rectangles,
//    each taken as a ball of radius 2^-20 (|tvec|_1 + |e1|_1 + |e2|_1) for o'_t vs o and the evaluation
//    of the points.  rho' = rho_t (1 + 1e-5) + that same slack.  Slivers (the pre-filter passes them
//    on for every ray) are global.
//  * (E_t): k_bin_triangles evaluates (P) per triangle (tri_escape, the function the group cones are
//    built from); a triangle the camera is nearly in the plane of leaves a cone entry (n_t, beta_t),
//    and k_bin_tri_escape appends it to every tile that holds a ray with |p . n_t| <= beta_t |p|: p . n_t
//    is affine in (s, t), so its range over the tile's (grown) rectangle is spanned by the corners,
//    and |p| is at most its largest corner value.
// Whatever cannot be listed -- a tile with more than kTileListCap primitives, more than
// kTileGlobalCap global ones, more than kTileEscCap cone entries, a singular camera frame, a band
// that does not start on a multiple of 4 rows -- takes the three-level sweep instead (the consumer
// returns false): lists only ever REPLACE the question "what may this tile's rays touch" and never
// a test.  The order of the tests is free: an equal closest t goes to the lower original index.
// Slots of a list past its count hold ids of earlier frames of the same scene or zeros -- valid
// slots either way, and testing a primitive twice or needlessly cannot change a closest hit -- so
// lists are read in whole batches of 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_brute.h"
#include "rt_device.h"
#include "rt_math.h"
#include "rt_tile_math.h"

namespace esc {

DEVINL void list_global(const TileLists &L, int id) {
  const int slot = atomicAdd(&L.hdr[0], 1);
  if (slot < kTileGlobalCap) L.hdr[8 + slot] = id;
}

// append `id` to every tile of the band that the pixel rectangle overlaps, the wave side by side;
// global when the rectangle spans very many tiles
DEVINL void list_rect(const RenderParams &p, const TileLists &L, int w0, int w1, int h0, int h1, int id,
                      int lane) {
  const int tx0 = w0 >> 5, tx1 = w1 >> 5, j0 = h0 >> 2, j1 = h1 >> 2;
  if ((long long)(tx1 - tx0 + 1) * (j1 - j0 + 1) > kTileMaxSpan) {
    if (lane == 0) list_global(L, id);
    return;
  }
  const int nx = tx1 - tx0 + 1, n = nx * (j1 - j0 + 1);
  for (int k = lane; k < n; k += 64) {
    const int r4 = band_tile_row(p, j0 + k / nx);
    if (r4 < 0) continue;
    const int tile = r4 * L.tiles_x + tx0 + k % nx;
    const int slot = atomicAdd(&L.cnt[tile], 1);
    if (slot < kTileListCap) L.ids[(size_t)tile * kTileListCap + slot] = id;
  }
}

// ... and the same for a CONVEX region given by its support along three directions as well: a tile
// none of whose pixels can lie inside [lo_a, hi_a] along some direction n_a is left out.  (The
// triangle's region is the convex hull of 12 small rectangles; its bounding box alone hands a large
// triangle seen at an angle to twice the tiles it touches.)  Tiles are taken one pixel larger all
// round, like the rectangle's own floor / ceil; a NaN anywhere compares false and keeps the tile.
struct HullAxes {
  double nx[3], ny[3], lo[3], hi[3];
};
DEVINL void list_rect_hull(const RenderParams &p, const TileLists &L, int w0, int w1, int h0, int h1, int id,
                           int lane, const HullAxes &A) {
  const int tx0 = w0 >> 5, tx1 = w1 >> 5, j0 = h0 >> 2, j1 = h1 >> 2;
  if ((long long)(tx1 - tx0 + 1) * (j1 - j0 + 1) > kTileMaxSpan) {
    if (lane == 0) list_global(L, id);
    return;
  }
  const int nx = tx1 - tx0 + 1, n = nx * (j1 - j0 + 1);
  for (int k = lane; k < n; k += 64) {
    const int j = j0 + k / nx, tx = tx0 + k % nx;
    const int r4 = band_tile_row(p, j);
    if (r4 < 0) continue;
    const double x0 = 32.0 * tx - 1.0, x1 = 32.0 * tx + 32.0, y0 = 4.0 * j - 1.0, y1 = 4.0 * j + 4.0;
    bool outside = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double ax0 = A.nx[a] * x0, ax1 = A.nx[a] * x1, ay0 = A.ny[a] * y0, ay1 = A.ny[a] * y1;
      const double tmin = fmin(ax0, ax1) + fmin(ay0, ay1), tmax = fmax(ax0, ax1) + fmax(ay0, ay1);
      outside |= (tmax < A.lo[a]) | (tmin > A.hi[a]);
    }
    if (outside) continue;
    const int tile = r4 * L.tiles_x + tx;
    const int slot = atomicAdd(&L.cnt[tile], 1);
    if (slot < kTileListCap) L.ids[(size_t)tile * kTileListCap + slot] = id;
  }
}

// one wave per slot of the group-sorted sphere table
__global__ void __launch_bounds__(256) k_bin_spheres(const RenderParams p) {
  const int s = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  if (s >= p.sg.n_grp * kSphGroup) return;
  const DevSph S = p.sg.sorted[s];
  if (!(S.r2 >= 0.f)) return; // pad slot (r2 = -inf): never hit
  const TileLists L = p.sl;
  const CamD cam = cam_frame(p);
  if (!cam.ok) {
    if (lane == 0) L.hdr[2] = 1;
    return;
  }
  const double c[3] = {(double)S.cx - cam.o[0], (double)S.cy - cam.o[1], (double)S.cz - cam.o[2]};
  const double A1 = fabs(c[0]) + fabs(c[1]) + fabs(c[2]);
  // D_i of rt_brute.h "Sphere GROUPS" (4.41 * 2^-12 < 0x1.2p-10), a little larger, + c~_i vs c_i
  const double R = (sqrt((double)S.r2) + 0x1.2p-10 * A1) * (1.0 + 0x1p-20) + 0x1p-22 * A1 + 0x1p-60;
  int w0, w1, h0, h1;
  const int st = sphere_pixel_rect(cam, p.W, p.H, c, R, w0, w1, h0, h1);
  if (st == 2) return; // off screen
  if (st == 0) {
    if (lane == 0) list_global(L, s);
    return;
  }
  list_rect(p, L, w0, w1, h0, h1, s, lane);
}

// (P) of rt_brute.h "Triangle GROUPS": may a ray from the camera o be accepted by triangle T through
// the pre-filter's "nearly parallel" escape at all?  Only if the camera lies within H of T's plane.
// Everything in fp32 from T's own record; 1 % + 16u |tvec| on top of H cover the roundings, and
// every doubtful case (a sliver, a non-finite value) answers yes with no usable normal.
struct TriEscape {
  bool possible; // the camera is within H of the plane (or nothing can be said)
  bool bounded;  // nh / beta / H below are valid: the escape needs |d . nh| < beta
  f3 nh;         // unit normal
  float beta;    // tau / |n1| for this origin distance
  float H;       // the origin must lie within H of the plane
  float U, V;    // an escape accept has |un*| <= U, |vn*| <= V (the exact numerators; see k_bin_triangles)
};
// (P) for ray origins with |origin - v0|_1 <= at: beta and H grow with at, so a bound on at gives a
// bound for every origin it covers.  `pad` out: the slot holds no triangle (all zeros: det == 0).
DEVINL TriEscape tri_escape_at(const DevTri &T, float at, float slack_k, bool &pad) {
  TriEscape E;
  E.possible = true;
  E.bounded = false;
  E.nh = mk(0.f, 0.f, 0.f);
  E.beta = 0.f;
  E.H = 0.f;
  E.U = E.V = __builtin_huge_valf();
  pad = false;
  const float u = 0x1p-24f;
  const f3 e1 = ld3(T.e1), e2 = ld3(T.e2);
  if (e1.x == 0.f && e1.y == 0.f && e1.z == 0.f && e2.x == 0.f && e2.y == 0.f && e2.z == 0.f) {
    E.possible = false; // pad slot (or a point): det == 0 exactly, rejected
    pad = true;
    return E;
  }
  const f3 n1 = cross(e2, e1);
  const float nn = sqrtf(dot(n1, n1));
  const float a1 = (fabsf(e1.x) + fabsf(e1.y)) + fabsf(e1.z);
  const float a2 = (fabsf(e2.x) + fabsf(e2.y)) + fabsf(e2.z);
  const float l1 = sqrtf(dot(e1, e1)), l2 = sqrtf(dot(e2, e2));
  const f3 s3 = (e1 + e2) * (1.f / 3.f);
  const float rho = sqrtf(fmaxf(fmaxf(dot(s3, s3), dot(e1 - s3, e1 - s3)), dot(e2 - s3, e2 - s3)));
  const float emax = fmaxf(l1, l2);
  if (!(rho > 0x1.2p-10f * emax) || !(nn > 0.f)) return E; // sliver / no normal
  const float p12 = a1 * a2;
  const float k = 3.2f * u * emax / rho * 1.0001f / slack_k; // this level's threshold tau / k
  const float tau = k * ((10.04f * a2 + 5.04f * a1) * at + 20.1f * p12);
  if (!(tau < 0.1f * nn)) return E; // |d . n| < 0.1 is part of the argument
  const float ted = (tau + 10.05f * u * p12) * (1.f + 4.f * u);
  const float U = ted + 10.04f * u * at * a2, V = ted + 5.04f * u * at * a1;
  E.H = (fmaxf(V / l1, U / l2) + at * tau / nn) * (2.f * l1 * l2 / nn) * (1.f / 0.99f);
  E.nh = n1 * (1.f / nn);
  E.beta = tau / nn;
  E.U = U;
  E.V = V;
  E.bounded = (E.nh.x == E.nh.x) && (E.nh.y == E.nh.y) && (E.nh.z == E.nh.z) && (E.beta == E.beta) &&
              (E.H == E.H);
  return E;
}
DEVINL TriEscape tri_escape(const DevTri &T, f3 o, float slack_k) {
  const f3 tv = o - ld3(T.v0);
  const float at = (fabsf(tv.x) + fabsf(tv.y)) + fabsf(tv.z);
  bool pad;
  TriEscape E = tri_escape_at(T, at, slack_k, pad);
  if (pad || !E.bounded) return E;
  const f3 n1 = cross(ld3(T.e2), ld3(T.e1));
  const float h = fabsf(dot(tv, n1)) / sqrtf(dot(n1, n1));
  E.possible = !(h > E.H * 1.01f + 0x1p-20f * at); // NaN: yes
  return E;
}


// one wave per slot of the group-sorted triangle table: global / (S_t) rectangle / (E_t) cone entry
__global__ void __launch_bounds__(256) k_bin_triangles(const RenderParams p) {
  const int k = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  if (k >= p.tg.n_grp * kTriGroup) return;
  const DevTri T = p.tg.sorted[k];
  const f3 e1f = ld3(T.e1), e2f = ld3(T.e2);
  if (e1f.x == 0.f && e1f.y == 0.f && e1f.z == 0.f && e2f.x == 0.f && e2f.y == 0.f && e2f.z == 0.f)
    return; // pad slot (or a point): det == 0 exactly, never accepted
  const TileLists L = p.tl;
  const CamD cam = cam_frame(p);
  if (!cam.ok) {
    if (lane == 0) L.hdr[2] = 1;
    return;
  }
  const double e1[3] = {e1f.x, e1f.y, e1f.z}, e2[3] = {e2f.x, e2f.y, e2f.z};
  const double v0[3] = {(double)T.v0[0] - cam.o[0], (double)T.v0[1] - cam.o[1], (double)T.v0[2] - cam.o[2]};
  double n1[3];
  cross3(e2, e1, n1);
  const double nn = sqrt(dot3(n1, n1)), l1 = sqrt(dot3(e1, e1)), l2 = sqrt(dot3(e2, e2));
  const double s3[3] = {(e1[0] + e2[0]) / 3.0, (e1[1] + e2[1]) / 3.0, (e1[2] + e2[2]) / 3.0};
  const double q1[3] = {e1[0] - s3[0], e1[1] - s3[1], e1[2] - s3[2]}, q2[3] = {e2[0] - s3[0], e2[1] - s3[1], e2[2] - s3[2]};
  const double rho = sqrt(fmax(fmax(dot3(s3, s3), dot3(q1, q1)), dot3(q2, q2)));
  const double emax = fmax(l1, l2);
  // (P) / (E_t) per triangle, with the function the group cones are built from
  const TriEscape E = tri_escape(T, mk(p.origin[0], p.origin[1], p.origin[2]), 1.f);
  const bool sliver = !(rho > 0x1.4p-10 * emax) || !(nn > 0.0) || !(l1 > 0.0); // (the pre-filter's own
                                                                                // threshold is 2^-10)
  if (sliver || (E.possible && !E.bounded)) { // passed on for every ray
    if (lane == 0) list_global(L, k);
    return;
  }
  // Statement (K) of rt_brute.h trades the two halves (as the triangle light lists do per light): with
  // the escape threshold tau_t / k an accepted hit lies within k rho_t of its triangle in its plane,
  // and the escape needs the camera within H_t(tau_t / k) of the plane.  Per camera each triangle takes
  // the SMALLEST k of 1/8, 1/4, 1/2, 1 at which this camera cannot take the escape at all: an eighth
  // of the dilation for nearly every triangle (the region is then the triangle itself, a little
  // grown, instead of a shape three times its size); a triangle the camera is nearly in the plane of
  // keeps k = 1 and leaves its entry for k_bin_tri_escape.
  double kK = 1.0;
  if (!E.possible) {
    for (int tr = 0; tr < 3; ++tr) {
      const float kc = 0.125f * (float)(1 << tr);
      const TriEscape Ec = tri_escape(T, mk(p.origin[0], p.origin[1], p.origin[2]), kc);
      if (Ec.bounded && !Ec.possible) {
        kK = (double)kc;
        break;
      }
    }
  }
  if (E.possible && lane == 0) { // the camera is nearly in this plane: k_bin_tri_escape bins the band
    const int slot = atomicAdd(&L.hdr[1], 1);
    if (slot < kTileEscCap) {
      TileEsc X;
      const double nh[3] = {E.nh.x, E.nh.y, E.nh.z};
      double A[3], Hh[3], V[3];
      for (int j = 0; j < 3; ++j) {
        A[j] = (double)p.llc[j] - cam.o[j];
        Hh[j] = p.horizontal[j];
        V[j] = p.vertical[j];
      }
      X.fA = dot3(A, nh);
      X.fH = dot3(Hh, nh);
      X.fV = dot3(V, nh);
      X.kp = E.beta * 1.0001f + 0x1p-20f; // fp32 beta and normal, |d| - 1
      X.id = k;
      // Escape rays pass the edges' planes too.  An accept through the escape has |det*| < tau, hence
      // |det_ref| <= tau + ed, and it needs 0 < un / det, vn / det <= 1 + 4u in the reference's own
      // numerators (rt_brute.h "Triangle pre-filter"): |un*| <= (tau + ed)(1 + 4u) + eu = U, |vn*| <= V
      // (tri_escape_at's U, V, built on |tvec|_1 |e|_1 >= the products the bounds are stated in).  The
      // exact numerators are triple products of the reference's direction d:
      //     un* = tvec . (d x e2) = d . (e2 x tvec),      vn* = d . (tvec x e1),
      // tvec = fl(o - v0) as the reference rounds it -- two more planes through the ray origin, each
      // holding an edge line, that d is nearly parallel to.  Their bands are affine in (s, t) like the
      // first; a tile gets the triangle only where all three cross it: the neighbourhood of the
      // triangle's own image instead of a stripe across the frame.
      const f3 tvf = mk(p.origin[0], p.origin[1], p.origin[2]) - ld3(T.v0);
      const double tvd[3] = {tvf.x, tvf.y, tvf.z};
      double mu[3], mv[3];
      cross3(e2, tvd, mu);
      cross3(tvd, e1, mv);
      const double lu = sqrt(dot3(mu, mu)), lv = sqrt(dot3(mv, mv));
      const double ku = (double)E.U / lu * 1.0001 + 0x1p-20, kv = (double)E.V / lv * 1.0001 + 0x1p-20;
      const bool use_u = lu > 0.0 && ku < 0.5, use_v = lv > 0.0 && kv < 0.5; // (NaN: no constraint)
      X.uA = use_u ? dot3(A, mu) / lu : 0.0;
      X.uH = use_u ? dot3(Hh, mu) / lu : 0.0;
      X.uV = use_u ? dot3(V, mu) / lu : 0.0;
      X.kpu = use_u ? (float)(ku * (1.0 + 0x1p-20)) : 2.f; // |p . 0| <= 2 |p|: every tile
      X.vA = use_v ? dot3(A, mv) / lv : 0.0;
      X.vH = use_v ? dot3(Hh, mv) / lv : 0.0;
      X.vV = use_v ? dot3(V, mv) / lv : 0.0;
      X.kpv = use_v ? (float)(kv * (1.0 + 0x1p-20)) : 2.f;
      X.pad[0] = X.pad[1] = 0;
      L.esc[slot] = X;
    }
  }
  // (S_t): lanes 0..11 take the corners v_i +- rho' e_a +- rho' e_b
  const double at = (fabs(v0[0]) + fabs(v0[1]) + fabs(v0[2])) + (fabs(e1[0]) + fabs(e1[1]) + fabs(e1[2])) +
                    (fabs(e2[0]) + fabs(e2[1]) + fabs(e2[2]));
  const double slack = 0x1p-20 * at + 0x1p-60;
  const double rp = kK * rho * (1.0 + 1e-5) + slack;
  const double ea[3] = {e1[0] / l1, e1[1] / l1, e1[2] / l1};
  double eb[3];
  cross3(n1, ea, eb);
  for (int j = 0; j < 3; ++j) eb[j] /= nn;
  const int cn = lane % 12, vi = cn >> 2;
  const double sa = (cn & 1) ? rp : -rp, sb = (cn & 2) ? rp : -rp;
  double pt[3];
  for (int j = 0; j < 3; ++j)
    pt[j] = v0[j] + (vi == 1 ? e1[j] : 0.0) + (vi == 2 ? e2[j] : 0.0) + sa * ea[j] + sb * eb[j];
  double ext[4], depth = 0.0;
  const bool bounded = sphere_pixel_extent(cam, p.W, p.H, pt, slack, ext, &depth);
  // a corner in the camera plane, or corners on both sides of it: the hull of the projections
  // says nothing (the triangle's image runs through infinity)
  const unsigned long long front = __builtin_amdgcn_ballot_w64(depth > 0.0);
  if (__builtin_amdgcn_ballot_w64(!bounded) != 0 || (front != 0 && front != ~0ull)) {
    if (lane == 0) list_global(L, k);
    return;
  }
  // the hull of the 12 corner rectangles along the normals of the projected triangle's three edges
  // (any direction would do for the argument; these are the ones that cut): vertex i's image is
  // taken as the mean of its four corners' centres (lanes 4 i .. 4 i + 3)
  const double ccx = 0.5 * (ext[0] + ext[1]), ccy = 0.5 * (ext[2] + ext[3]);
  const double chx = 0.5 * (ext[1] - ext[0]), chy = 0.5 * (ext[3] - ext[2]);
  double vx[3], vy[3];
  for (int i = 0; i < 3; ++i) {
    vx[i] = 0.25 * (__shfl(ccx, 4 * i) + __shfl(ccx, 4 * i + 1) + __shfl(ccx, 4 * i + 2) + __shfl(ccx, 4 * i + 3));
    vy[i] = 0.25 * (__shfl(ccy, 4 * i) + __shfl(ccy, 4 * i + 1) + __shfl(ccy, 4 * i + 2) + __shfl(ccy, 4 * i + 3));
  }
  HullAxes HA;
  for (int a = 0; a < 3; ++a) {
    const int b = (a + 1) % 3;
    HA.nx[a] = vy[b] - vy[a];
    HA.ny[a] = vx[a] - vx[b];
    const double c = HA.nx[a] * ccx + HA.ny[a] * ccy, r = fabs(HA.nx[a]) * chx + fabs(HA.ny[a]) * chy;
    double lo = c - r, hi = c + r;
    for (int off = 32; off > 0; off >>= 1) {
      lo = fmin(lo, __shfl_xor(lo, off));
      hi = fmax(hi, __shfl_xor(hi, off));
    }
    const double pad = 1e-6 * (fabs(lo) + fabs(hi)) + 1e-6; // the arithmetic of this test itself
    HA.lo[a] = lo - pad;
    HA.hi[a] = hi + pad;
  }
  for (int off = 32; off > 0; off >>= 1) { // union over the lanes (12 distinct corners, repeated)
    ext[0] = fmin(ext[0], __shfl_xor(ext[0], off));
    ext[1] = fmax(ext[1], __shfl_xor(ext[1], off));
    ext[2] = fmin(ext[2], __shfl_xor(ext[2], off));
    ext[3] = fmax(ext[3], __shfl_xor(ext[3], off));
  }
  int w0, w1, h0, h1;
  if (extent_rect(ext, p.W, p.H, w0, w1, h0, h1) == 2) return; // off screen
  list_rect_hull(p, L, w0, w1, h0, h1, k, lane, HA);
}

// (E_t): one workgroup per tile row: p . n = fA + s fH + t fV is affine, so over the row's t range the band
// |p . n| <= kp |p| is an interval of s (band_row_tiles); the candidate tiles then pass the per-tile
// test with their own largest |p|.  What depends on the row and the tile only -- the row's t range,
// every tile's pmax (four double square roots) -- is computed once per workgroup into LDS, the
// threads then take the frame's entries side by side.  (One thread per tile against every entry
// cost 1.3 ms per camera position on c5, one thread per (entry, row) 0.83 ms.)
constexpr int kEscRowChunk = 1024; // tiles of a row per pass (W <= 32,768 in one)
__global__ void __launch_bounds__(256) k_bin_tri_escape(const RenderParams p) {
  __shared__ double s_pmax[kEscRowChunk];
  const TileLists L = p.tl;
  const int r4 = (int)blockIdx.x, tid = (int)threadIdx.x;
  const int n_esc = L.hdr[1];
  if (n_esc > kTileEscCap) {
    if (r4 == 0 && tid == 0) L.hdr[2] = 1; // too many cones: the sweep handles this frame
    return;
  }
  if (n_esc == 0) return;
  const CamD cam = cam_frame(p);
  if (!cam.ok) return; // (hdr[2] already set by k_bin_triangles)
  const int h = band_image_row(p, 4 * r4);
  double st[4], pm_l, pm_r;
  tile_st_rect(p, cam, 0, h, st, pm_l);
  const double t0 = st[2], t1 = st[3];
  tile_st_rect(p, cam, L.tiles_x - 1, h, st, pm_r);
  const double pmax_row = fmax(pm_l, pm_r);
  for (int c0 = 0; c0 < L.tiles_x; c0 += kEscRowChunk) {
    const int c1 = min(L.tiles_x, c0 + kEscRowChunk);
    __syncthreads();
    for (int tx = c0 + tid; tx < c1; tx += 256) {
      tile_st(p, cam, tx, h, st);
      s_pmax[tx - c0] = tile_pmax(p, cam, st);
    }
    __syncthreads();
    for (int k = tid; k < n_esc; k += 256) {
      const TileEsc X = L.esc[k];
      int tx0, tx1, ux0, ux1, vx0, vx1;
      if (!band_row_tiles_with(p, cam, L.tiles_x, pmax_row, t0, t1, X.fA, X.fH, X.fV, (double)X.kp, tx0, tx1) ||
          !band_row_tiles_with(p, cam, L.tiles_x, pmax_row, t0, t1, X.uA, X.uH, X.uV, (double)X.kpu, ux0, ux1) ||
          !band_row_tiles_with(p, cam, L.tiles_x, pmax_row, t0, t1, X.vA, X.vH, X.vV, (double)X.kpv, vx0, vx1)) {
        L.hdr[2] = 1; // nothing can be said
        continue;
      }
      tx0 = max(tx0, max(ux0, vx0)); // the three bands' tiles of this row
      tx1 = min(tx1, min(ux1, vx1));
      for (int tx = max(tx0, c0); tx <= min(tx1, c1 - 1); ++tx) {
        tile_st(p, cam, tx, h, st);
        const double pm = s_pmax[tx - c0];
        if (!tile_band_hit(st, pm, X.fA, X.fH, X.fV, (double)X.kp) ||
            !tile_band_hit(st, pm, X.uA, X.uH, X.uV, (double)X.kpu) ||
            !tile_band_hit(st, pm, X.vA, X.vH, X.vV, (double)X.kpv))
          continue;
        const int tile = r4 * L.tiles_x + tx;
        const int slot = atomicAdd(&L.cnt[tile], 1);
        if (slot < kTileListCap) L.ids[(size_t)tile * kTileListCap + slot] = X.id;
      }
    }
  }
}

// ---- the list-driven primary sweep.  `tile` is wave-uniform.  false: this tile has no usable list.
typedef const int32_t __attribute__((address_space(4))) *ListPtr;

// body4(i0, i1, i2, i3): test four listed slots
template <typename Body4> DEVINL bool sweep_tile_list(const TileLists &L, int tile, Body4 body4) {
  const ListPtr hdr = (ListPtr)(uintptr_t)L.hdr;
  const int n_glob = hdr[0];
  if (hdr[2] != 0 || n_glob > kTileGlobalCap) return false;
  const int n = ((ListPtr)(uintptr_t)L.cnt)[tile];
  if (n > kTileListCap) return false;
  const SmemFetch<DevIdx4> glob{reinterpret_cast<const DevIdx4 *>(L.hdr + 8)};
  for (int k = 0; k < n_glob; k += 4) {
    const DevIdx4 I = glob(k >> 2);
    body4(I.v[0], I.v[1], I.v[2], I.v[3]);
  }
  const SmemFetch<DevIdx4> ids{reinterpret_cast<const DevIdx4 *>(L.ids + (size_t)tile * kTileListCap)};
  for (int k = 0; k < n; k += 4) {
    const DevIdx4 I = ids(k >> 2);
    body4(I.v[0], I.v[1], I.v[2], I.v[3]);
  }
  return true;
}

// four listed spheres (slots of the sorted table): the reference arithmetic, ties to the lower
// original index
template <typename FetchE>
DEVINL void sph4_listed_primary(FetchE rece, const int32_t *orig, int i0, int i1, int i2, int i3, int base,
                                const V3<v2f> &d, Hit (&h)[2]) {
  const ListPtr og = (ListPtr)(uintptr_t)orig;
  const DevSphP E[4] = {rece(i0), rece(i1), rece(i2), rece(i3)};
  DevIdx4 O;
  O.v[0] = og[i0];
  O.v[1] = og[i1];
  O.v[2] = og[i2];
  O.v[3] = og[i3];
  test_sph_primary_sorted(E, O, base, d, h);
}

// two listed triangles: the triangle filter, then the reference arithmetic
template <typename FetchF, typename FetchE>
DEVINL void tri2_listed_primary(FetchF recf, FetchE rece, const int32_t *orig, int k0, int k1,
                                const V3<v2f> &d, Hit (&h)[2]) {
  const TriF T[2] = {recf(k0), recf(k1)};
  v2f A[2], B[2], C[2];
  tri2_primary_filter_pk(T, d.x, d.y, d.z, A, B, C);
  const int m = tri_flags(A[1], B[1], C[1], tri_flags(A[0], B[0], C[0], -1));
  if (ANY_LANE_RARE(m >= 0)) {
    const ListPtr og = (ListPtr)(uintptr_t)orig;
    const DevTriP E[2] = {rece(k0), rece(k1)};
    test_tri2_primary_sorted(E, og[k0], og[k1], d, h);
  }
}


// ---------------------------------------------------------------------------------------
// Light lists (rt_device.h LightLists).  For a shadow ray (O, L) towards the sample point P that
// starts inside the (grown) scene box B -- rays from outside take the sweep -- rt_brute.h "Sphere
// GROUPS for shadow rays" gives: sphere i not rejected at `disc < 0`  ==>  the ray's LINE passes within
//     r_i (1+u) + 5.4 sqrt(u) |O - c_i| + 2^-74  <=  r_i (1+u) + 0x1.6p-10 W_i,  W_i = max over B's corners |x - c_i|
// of c_i.  L = fl(normalize(fl(P - O))) is within 3u of the true direction, so the line also passes
// within delta = 2^-20 (|B|_1 + |P - g|_1) of P; shifted by that much it runs through P itself
// and within R_i = (the reach above) + delta of c_i: the direction from P to the ray's origin lies in
// the disc of angular radius asin(R_i / |c_i - P|) around c_i - P.  An accept also needs t2 < |P - O|
// - eps, i.e. the sphere on the ray's side of P -- unless P is (nearly) inside the reach, and such
// spheres are listed for every direction.  The disc is binned per cube face with the conic bounds
// of the tile lists (rt_tile_math.h, the face taken as a camera at P) when the sphere lies wholly
// in front of the face's plane through P; a sphere that plane cuts can only matter to the face if
// its disc reaches within 54.74 degrees (the face's corner) of the face's axis, and then goes on the
// face's own short list.  The look-up (light_cell) uses fl(O - P), which is exactly -fl(P - O): within
// 3u of the line's direction, 3e-5 cells; rectangles are grown by 1e-3 cells.
// Lists hold PAIR records (two neighbours of the spatial order): the reference arithmetic on
// pair-interleaved records is what the sweeps' exact path runs (pair2_any_pk).
// ---------------------------------------------------------------------------------------
// one wave per pair record of the group-sorted table; every listed light, every face
__global__ void __launch_bounds__(256) k_bin_light_pairs(const RenderParams p) {
  const int j = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  const int n_rec = p.sg.n_grp * (kSphGroup / 2);
  if (j >= n_rec) return;
  const LightLists LL = p.ll;
  const DevSphPair S = p.sg.sorted2[j];
  const double g[3] = {p.shadow_center[0], p.shadow_center[1], p.shadow_center[2]};
  const double rho = (double)p.shadow_rho_max;
  const int Rr = LL.R, cells_per_face = Rr * Rr;
  for (int li = 0; li < LL.n_listed; ++li) {
    const double P[3] = {p.light_points[4 * LL.point[li]], p.light_points[4 * LL.point[li] + 1],
                         p.light_points[4 * LL.point[li] + 2]};
    const double delta = 0x1p-20 * (rho + fabs(P[0] - g[0]) + fabs(P[1] - g[1]) + fabs(P[2] - g[2]));
    double c[2][3], Rh[2], cn[2];
    bool real[2], around[2];
    for (int h = 0; h < 2; ++h) {
      real[h] = S.r2[h] >= 0.f; // pad half: r2 = -inf
      const double C[3] = {S.cx[h], S.cy[h], S.cz[h]};
      for (int k = 0; k < 3; ++k) c[h][k] = C[k] - P[k];
      double far2 = 0.0; // the farthest corner of the box of ray origins
      for (int k = 0; k < 3; ++k) {
        const double d = fmax(fabs(C[k] - (double)p.scene_lo[k]), fabs(C[k] - (double)p.scene_hi[k]));
        far2 += d * d;
      }
      const double reach = (sqrt(fmax(0.0, (double)S.r2[h])) + 0x1.6p-10 * sqrt(far2)) * (1.0 + 0x1p-20);
      Rh[h] = reach + delta + 0x1p-60;
      cn[h] = sqrt(dot3(c[h], c[h]));
      // P inside (or all but inside) the reach: every direction, and the "before P" argument is off
      around[h] = real[h] && !(cn[h] > Rh[h] * 1.001 + 1e-4 * rho);
    }
    for (int face = 0; face < 6; ++face) {
      int32_t *hdr = LL.hdr + (size_t)(li * 6 + face) * kTileHdrInts;
      int m, ia, ib;
      double sign;
      light_face_axes(face, m, ia, ib, sign);
      bool face_global = false, have = false;
      double ext[4] = {1e300, -1e300, 1e300, -1e300};
      for (int h = 0; h < 2; ++h) {
        if (!real[h]) continue;
        if (around[h]) {
          face_global = true;
          continue;
        }
        const double depth = sign * c[h][m];
        if (depth > Rh[h] * (1.0 + 1e-9)) { // wholly in front of the face's plane through P
          const CamD cam = light_face_frame(P, face);
          double e[4];
          if (!sphere_pixel_extent(cam, Rr + 1, Rr + 1, c[h], Rh[h], e, nullptr, 1e-3)) {
            face_global = true;
            continue;
          }
          ext[0] = fmin(ext[0], e[0]);
          ext[1] = fmax(ext[1], e[1]);
          ext[2] = fmin(ext[2], e[2]);
          ext[3] = fmax(ext[3], e[3]);
          have = true;
        } else {
          // cut by (or behind) the plane: relevant only if the disc reaches the face's directions,
          // all within acos(1 / sqrt 3) = 0.95532 rad of the axis
          const double ang = acos(fmax(-1.0, fmin(1.0, depth / cn[h])));
          const double phi = asin(fmin(1.0, Rh[h] / cn[h]));
          if (!(ang > 0.95532 + phi + 1e-6)) face_global = true;
        }
      }
      if (face_global) {
        if (lane == 0) {
          const int slot = atomicAdd(&hdr[0], 1);
          if (slot < kTileGlobalCap) hdr[8 + slot] = j;
        }
        continue;
      }
      if (!have) continue;
      const int u0 = max(0, (int)floor(ext[0])), u1 = min(Rr - 1, (int)floor(ext[1]));
      const int w0 = max(0, (int)floor(ext[2])), w1 = min(Rr - 1, (int)floor(ext[3]));
      if (u0 > u1 || w0 > w1) continue; // seen from P through other faces only
      const int nx = u1 - u0 + 1, n = nx * (w1 - w0 + 1);
      const size_t cell0 = (size_t)(li * 6 + face) * cells_per_face;
      for (int k = lane; k < n; k += 64) {
        const size_t cell = cell0 + (size_t)(w0 + k / nx) * Rr + u0 + k % nx;
        const int slot = atomicAdd(&LL.cnt[cell], 1);
        if (slot < kLightListCap) LL.ids[cell * kLightListCap + slot] = j;
      }
    }
  }
}

// The last light's sweep stops at ANY occluder (liberty 5: the order of its tests cannot be observed),
// and a wave stops as soon as none of its rays is still looking: each cell's records are put in the
// order of the solid angle they fill seen from P -- (r / |c - P|)^2 of the larger half -- so that a
// ray in shadow meets its most likely occluder first.  One thread per cell, an insertion sort of at
// most kLightListCap entries.  Nothing but the order changes (ties: the lower record first, which also
// makes the lists independent of the order the atomics of the binning kernels happened to run in).
template <bool TRI> __global__ void __launch_bounds__(256) k_sort_light_cells(const RenderParams p) {
  const LightLists LL = TRI ? p.lt : p.ll;
  const int cells_per_light = 6 * LL.R * LL.R;
  const int cell = blockIdx.x * 256 + (int)threadIdx.x;
  if (cell >= LL.n_listed * cells_per_light) return;
  const int n = LL.cnt[cell];
  if (n < 2 || n > kLightListCap) return;
  const int li = cell / cells_per_light;
  const float P[3] = {p.light_points[4 * LL.point[li]], p.light_points[4 * LL.point[li] + 1],
                      p.light_points[4 * LL.point[li] + 2]};
  auto key_of = [&](int j) -> float {
    float best = 0.f;
    for (int h = 0; h < 2; ++h) {
      float r2, d2;
      if (TRI) {
        const DevTri T = p.tg.sorted[2 * j + h];
        const float c[3] = {T.v0[0] + (T.e1[0] + T.e2[0]) / 3.f - P[0], T.v0[1] + (T.e1[1] + T.e2[1]) / 3.f - P[1],
                            T.v0[2] + (T.e1[2] + T.e2[2]) / 3.f - P[2]};
        const float x[3] = {T.e1[1] * T.e2[2] - T.e1[2] * T.e2[1], T.e1[2] * T.e2[0] - T.e1[0] * T.e2[2],
                            T.e1[0] * T.e2[1] - T.e1[1] * T.e2[0]};
        r2 = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); // twice the area
        d2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      } else {
        const DevSphPair S = p.sg.sorted2[j];
        const float c[3] = {S.cx[h] - P[0], S.cy[h] - P[1], S.cz[h] - P[2]};
        r2 = S.r2[h]; // a pad half: -inf
        d2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      }
      const float k = r2 / fmaxf(d2, 1e-30f);
      if (k > best) best = k; // (NaN: not larger)
    }
    return best;
  };
  int32_t *ids = LL.ids + (size_t)cell * kLightListCap;
  int32_t v[kLightListCap];
  float key[kLightListCap];
  for (int i = 0; i < n; ++i) {
    const int32_t id = ids[i];
    const float k = key_of(id);
    int q = i;
    while (q > 0 && (key[q - 1] < k || (key[q - 1] == k && v[q - 1] > id))) {
      key[q] = key[q - 1];
      v[q] = v[q - 1];
      --q;
    }
    key[q] = k;
    v[q] = id;
  }
  for (int i = 0; i < n; ++i) ids[i] = v[i];
}

// the cell (over all listed lights and faces) of the shadow ray that starts at `o` towards the
// sample point P of listed light li
DEVINL int light_list_cell(const LightLists &LL, int li, f3 P, f3 o) {
  const f3 v = o - P;
  const float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z);
  int m = 0;
  float dm = ax;
  if (ay > dm) { m = 1; dm = ay; }
  if (az > dm) { m = 2; dm = az; }
  const float vm = (m == 0) ? v.x : (m == 1) ? v.y : v.z;
  const float va = (m == 0) ? v.y : v.x;
  const float vb = (m == 2) ? v.y : v.z;
  const int face = 2 * m + ((vm < 0.f) ? 1 : 0);
  const float hR = 0.5f * (float)LL.R;
  // (v_rcp_f32 is within 1 ulp: the quotients are off by < 2u, 3e-5 cells at R = 128 -- the rectangles
  // are grown by 1e-3 cells; nothing here reaches the image)
  const float rd = __builtin_amdgcn_rcpf(dm);
  const int cu = min(LL.R - 1, max(0, (int)floorf((va * rd + 1.f) * hR)));
  const int cw = min(LL.R - 1, max(0, (int)floorf((vb * rd + 1.f) * hR)));
  return ((li * 6 + face) * LL.R + cw) * LL.R + cu; // (NaN direction: cell 0 of some face; such a ray has tb = 0)
}

// Shadow rays of a wave through the light lists of light li: one distinct cell of its live rays at a
// time (wave-uniform: the records come through the scalar cache like everywhere else), every live
// ray tested against every list swept -- a ray's own cell holds all its candidates, the others'
// cannot add a wrong one.  Returns false when some live ray cannot be served (its cell or face list
// overflowed): the caller runs the group sweep for the wave, which is complete on its own.
// n_tests: per lane, sphere tests executed while the ray was live.
template <typename FetchE>
DEVINL bool anyhit_sph_light_lists(const LightLists &LL, int cell, FetchE rece, int base, f3 o, f3 L, Any &a,
                                   int &n_tests, int &swept) {
  const v2f oxy = {o.x, o.y}, oz_ = {o.z, 0.f}, Lxy = {L.x, L.y}, Lz_ = {L.z, 0.f};
  auto exact2 = [&](int k0, int k1) { // sorted pair records k0, k1: the reference arithmetic
    const PairG R[2] = {rece(k0), rece(k1)};
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
          any_accept(a, base + 2 * (i ? k1 : k0) + c, t2); // a position in the sorted table
      }
  };
  const int cells_per_face = LL.R * LL.R;
  const ListPtr cnt = (ListPtr)(uintptr_t)LL.cnt;
  unsigned long long done = 0;
  for (;;) {
    const unsigned long long todo = __builtin_amdgcn_ballot_w64(a.tb > 0.f) & ~done;
    if (todo == 0) return true;
    const int leader = __builtin_ctzll(todo);
    const int c = __builtin_amdgcn_readlane(cell, leader);
    const ListPtr hdr = (ListPtr)(uintptr_t)(LL.hdr + (size_t)(c / cells_per_face) * kTileHdrInts);
    const int n_glob = hdr[0], n = cnt[c];
    if (n_glob > kTileGlobalCap || n > kLightListCap || hdr[2] != 0) return false;
    const SmemFetch<DevIdx4> glob{reinterpret_cast<const DevIdx4 *>(LL.hdr + (size_t)(c / cells_per_face) * kTileHdrInts + 8)};
    const SmemFetch<DevIdx4> ids{reinterpret_cast<const DevIdx4 *>(LL.ids + (size_t)c * kLightListCap)};
    auto batch = [&](const DevIdx4 &I) { // four pair records = 8 spheres
      n_tests += (a.tb > 0.f) ? 8 : 0;
      swept += 8;
      exact2(I.v[0], I.v[1]);
      exact2(I.v[2], I.v[3]);
    };
    for (int k = 0; k < n_glob; k += 4) { // (lists are read in whole batches of 4: spare slots hold
      if (__builtin_amdgcn_ballot_w64(a.tb > 0.f) == 0) return true; //  valid records)
      batch(glob(k >> 2));
    }
    for (int k = 0; k < n; k += 4) {
      if (__builtin_amdgcn_ballot_w64(a.tb > 0.f) == 0) return true;
      batch(ids(k >> 2));
    }
    done |= __builtin_amdgcn_ballot_w64(cell == c);
  }
}


// ---------------------------------------------------------------------------------------
// Light lists of TRIANGLE pair records.  For a shadow ray (O, L) towards P that starts inside the
// grown scene box B (|O - v0|_1 <= at_t := the largest such 1-norm over B's corners), the pre-filter's
// statement with tau evaluated at at_t (rt_brute.h "Triangle pre-filter", rt_capi.cpp build_tri2pf):
//   triangle t accepted  ==>  (S_t) the ray's line meets t's plane at X* within k rho_t of t in that plane
//                             or (E_t) |L . n_t| < beta_t := tau_t(at_t) / (k |n1_t|), and then (P) O lies
//                             within H_t(at_t) of the plane (tri_escape_at: both grow with at); k: (K) below.
//  * (S_t) is binned like the tile lists' triangles: the hull of the 12 corners v_i +- rho' e_a +- rho' e_b,
//    each a ball of radius 2^-20 (at_t + |e1|_1 + |e2|_1) + delta (delta: the line misses P by at most
//    that, "Light lists" above), projected on each cube face.  TWO-sided: t2 of a nearly parallel ray
//    can be off by a sixth, so a triangle just beyond P cannot be ruled out by `t2 < |P - O|`; the
//    double cone of the conic bounds does that by itself (a corner behind the face's plane is
//    projected through P).  Corners on both sides of a face's plane, or in it: the face's short
//    list if the triangle's bounding ball reaches the face's directions at all, else nothing.
//  * (E_t): a ray towards P that is nearly parallel to t's plane and starts within H_t of it has P
//    within H_t + |P - O| beta_t of the plane.  Per triangle and light that is one comparison; the
//    few triangles that pass leave an entry (n_t, beta_t) and k_bin_light_tri_escape appends them to
//    every cell that holds a direction v with |v . n_t| <= beta_t |v| (v . n_t is affine over a cell).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bin_light_tri_pairs(const RenderParams p) {
  const int j = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  const int n_rec = p.tg.n_grp * (kTriGroup / 2);
  if (j >= n_rec) return;
  const LightLists LL = p.lt;
  const double g[3] = {p.shadow_center[0], p.shadow_center[1], p.shadow_center[2]};
  const double rho_max = (double)p.shadow_rho_max;
  const int Rr = LL.R, cells_per_face = Rr * Rr;
  // this lane's corner: triangle h of the pair, corner cn of its 12
  const int h = (lane % 24) / 12, cn = lane % 12, vi = cn >> 2;
  const DevTri T = p.tg.sorted[2 * j + h];
  const f3 e1f = ld3(T.e1), e2f = ld3(T.e2);
  const bool real = !(e1f.x == 0.f && e1f.y == 0.f && e1f.z == 0.f && e2f.x == 0.f && e2f.y == 0.f && e2f.z == 0.f);
  const double e1[3] = {e1f.x, e1f.y, e1f.z}, e2[3] = {e2f.x, e2f.y, e2f.z};
  const double v0[3] = {T.v0[0], T.v0[1], T.v0[2]};
  double n1[3];
  cross3(e2, e1, n1);
  const double nn = sqrt(dot3(n1, n1)), l1 = sqrt(dot3(e1, e1)), l2 = sqrt(dot3(e2, e2));
  const double s3[3] = {(e1[0] + e2[0]) / 3.0, (e1[1] + e2[1]) / 3.0, (e1[2] + e2[2]) / 3.0};
  const double q1[3] = {e1[0] - s3[0], e1[1] - s3[1], e1[2] - s3[2]}, q2[3] = {e2[0] - s3[0], e2[1] - s3[1], e2[2] - s3[2]};
  const double rho = sqrt(fmax(fmax(dot3(s3, s3), dot3(q1, q1)), dot3(q2, q2)));
  const bool sliver = real && (!(rho > 0x1.4p-10 * fmax(l1, l2)) || !(nn > 0.0) || !(l1 > 0.0));
  // at_t: the largest |O - v0|_1 over the box of ray origins
  double at = 0.0;
  for (int k = 0; k < 3; ++k)
    at += fmax(fabs(v0[k] - (double)p.scene_lo[k]), fabs(v0[k] - (double)p.scene_hi[k]));
  const double a12 = (fabs(e1[0]) + fabs(e1[1]) + fabs(e1[2])) + (fabs(e2[0]) + fabs(e2[1]) + fabs(e2[2]));
  // Statement (K) of rt_brute.h trades the two halves: with the escape threshold tau_t / k an accepted
  // hit lies within 0.9375 k rho_t + 4u emax <= k rho_t of its triangle (k >= 64u emax / rho_t, i.e.
  // k >= 0.004 for anything that is not a sliver), and the escape needs the origin within H_t(tau_t / k)
  // of the plane.  Each triangle takes, per light, the SMALLEST k of 1/8, 1/4, 1/2, 1 for which no ray
  // towards that light can take the escape at all (P further than H + Lmax beta from the plane): an
  // eighth of the dilation for nearly every triangle; a triangle that fails even at k = 1 (the
  // light sits in its plane) leaves an escape entry at k = 1.
  for (int li = 0; li < LL.n_listed; ++li) {
    const double P[3] = {p.light_points[4 * LL.point[li]], p.light_points[4 * LL.point[li] + 1],
                         p.light_points[4 * LL.point[li] + 2]};
    const double delta = 0x1p-20 * (rho_max + fabs(P[0] - g[0]) + fabs(P[1] - g[1]) + fabs(P[2] - g[2]));
    const double slack = 0x1p-20 * (at + a12) + delta + 0x1p-60;
    double c0[3], ctr[3]; // v0 and the centroid relative to P
    for (int k = 0; k < 3; ++k) {
      c0[k] = v0[k] - P[k];
      ctr[k] = c0[k] + s3[k];
    }
    double lmax2 = 0.0; // the longest ray towards P from inside the box
    for (int k = 0; k < 3; ++k) {
      const double d = fmax(fabs(P[k] - (double)p.scene_lo[k]), fabs(P[k] - (double)p.scene_hi[k]));
      lmax2 += d * d;
    }
    const double hP = real && nn > 0.0 ? fabs(dot3(c0, n1)) / nn : 0.0;
    double kK = 1.0;
    bool pad_slot, flagged = true;
    TriEscape E = tri_escape_at(T, (float)(at * (1.0 + 0x1p-20)), 1.f, pad_slot);
    double kp = (double)E.beta * 1.0001 + 0x1p-20;
    for (int tr = 0; tr < 4 && real; ++tr) {
      const double kc = 0.125 * (double)(1 << tr);
      const TriEscape Ec = tri_escape_at(T, (float)(at * (1.0 + 0x1p-20)), (float)kc, pad_slot);
      const double kpc = (double)Ec.beta * 1.0001 + 0x1p-20;
      const bool fl = !Ec.bounded || !(hP > (double)Ec.H * 1.01 + sqrt(lmax2) * kpc + slack);
      if (!fl || tr == 3) {
        kK = kc;
        E = Ec;
        kp = kpc;
        flagged = fl;
        break;
      }
    }
    const double rp = kK * rho * (1.0 + 1e-5) + slack;
    const double cdist = sqrt(dot3(ctr, ctr)), ball = rho + 1.5 * rp; // holds the 12 corners' balls
    // a pair with a sliver, a triangle without usable bounds, or P (all but) inside a bounding ball:
    // every direction of every face
    const bool everywhere = real && (sliver || (E.possible && !E.bounded) || !(cdist > ball * 1.001));
    // (E_t) + (P): can ANY ray towards P from inside the box take the escape?  P within H + Lmax beta
    if (real && !everywhere && E.bounded && cn == 0 && lane < 24) {
      if (flagged) {
        int32_t *hdr0 = LL.hdr + (size_t)(li * 6) * kTileHdrInts;
        const int slot = atomicAdd(&hdr0[1], 1);
        if (slot < kLightEscCap) {
          LightEsc X;
          X.nx = E.nh.x;
          X.ny = E.nh.y;
          X.nz = E.nh.z;
          X.kp = (float)kp;
          X.pair = j;
          X.pad[0] = X.pad[1] = X.pad[2] = 0;
          LL.esc[(size_t)li * kLightEscCap + slot] = X;
        }
      }
    }
    double ea[3] = {0, 0, 0}, eb[3] = {0, 0, 0};
    if (real && !sliver) {
      for (int k = 0; k < 3; ++k) ea[k] = e1[k] / l1;
      cross3(n1, ea, eb);
      for (int k = 0; k < 3; ++k) eb[k] /= nn;
    }
    const double sa = (cn & 1) ? rp : -rp, sb = (cn & 2) ? rp : -rp;
    double pt[3];
    for (int k = 0; k < 3; ++k)
      pt[k] = c0[k] + (vi == 1 ? e1[k] : 0.0) + (vi == 2 ? e2[k] : 0.0) + sa * ea[k] + sb * eb[k];
    const unsigned long long any_everywhere = __builtin_amdgcn_ballot_w64(everywhere);
    for (int face = 0; face < 6; ++face) {
      int32_t *hdr = LL.hdr + (size_t)(li * 6 + face) * kTileHdrInts;
      int m, ia, ib;
      double sign;
      light_face_axes(face, m, ia, ib, sign);
      bool face_global = any_everywhere != 0;
      int rect[2][4]; // per triangle of the pair: u0, u1, w0, w1 (cells), u0 > u1: none
      rect[0][0] = rect[1][0] = 1;
      rect[0][1] = rect[1][1] = 0;
      if (!face_global) {
        const CamD cam = light_face_frame(P, face);
        double e[4], depth = 0.0;
        const bool bounded = sphere_pixel_extent(cam, Rr + 1, Rr + 1, pt, slack, e, &depth, 1e-3);
        // per triangle of the pair: all 12 corners bounded and on one side of the face's plane?
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned long long mine = 0xFFFull << (12 * hh); // lanes 0..11 / 12..23 hold its corners
          const unsigned long long is_real = __builtin_amdgcn_ballot_w64(real) & mine;
          if (!is_real) continue;
          const unsigned long long nb = __builtin_amdgcn_ballot_w64(!bounded) & mine;
          const unsigned long long fr = __builtin_amdgcn_ballot_w64(depth > 0.0) & mine;
          if (nb == 0 && (fr == 0 || fr == mine)) {
            // union of this triangle's corner extents (lanes outside `mine` contribute nothing)
            const bool in = (lane < 24) && (lane / 12 == hh);
            double x0 = in ? e[0] : 1e300, x1 = in ? e[1] : -1e300, y0 = in ? e[2] : 1e300, y1 = in ? e[3] : -1e300;
            for (int off = 32; off > 0; off >>= 1) {
              x0 = fmin(x0, __shfl_xor(x0, off));
              x1 = fmax(x1, __shfl_xor(x1, off));
              y0 = fmin(y0, __shfl_xor(y0, off));
              y1 = fmax(y1, __shfl_xor(y1, off));
            }
            rect[hh][0] = max(0, (int)floor(x0));
            rect[hh][1] = min(Rr - 1, (int)floor(x1));
            rect[hh][2] = max(0, (int)floor(y0));
            rect[hh][3] = min(Rr - 1, (int)floor(y1));
            if (rect[hh][2] > rect[hh][3]) rect[hh][0] = 1, rect[hh][1] = 0; // seen through other faces only
          } else {
            // cut by the face's plane: relevant only if its bounding ball reaches the face's
            // directions (within acos(1 / sqrt 3) of the axis), seen from either side of P
            const int src = 12 * hh; // a lane that holds this triangle's numbers
            const double cd = __shfl(cdist, src), bl = __shfl(ball, src), cm = __shfl(ctr[m], src);
            const double ang = acos(fmax(-1.0, fmin(1.0, fabs(cm) / cd)));
            const double phi = asin(fmin(1.0, bl / cd));
            if (!(ang > 0.95532 + phi + 1e-6)) face_global = true;
          }
        }
      }
      if (face_global) {
        if (lane == 0) {
          const int slot = atomicAdd(&hdr[0], 1);
          if (slot < kTileGlobalCap) hdr[8 + slot] = j;
        }
        continue;
      }
      // the two triangles' rectangles: one append when they touch (neighbours in space mostly do),
      // two when they lie apart (one in front of the face's plane, one behind it: their images are
      // at opposite ends of the face, and a union would cover everything in between)
      const bool v0r = rect[0][0] <= rect[0][1], v1r = rect[1][0] <= rect[1][1];
      if (v0r && v1r && rect[0][0] <= rect[1][1] + 1 && rect[1][0] <= rect[0][1] + 1 &&
          rect[0][2] <= rect[1][3] + 1 && rect[1][2] <= rect[0][3] + 1) {
        rect[0][0] = min(rect[0][0], rect[1][0]);
        rect[0][1] = max(rect[0][1], rect[1][1]);
        rect[0][2] = min(rect[0][2], rect[1][2]);
        rect[0][3] = max(rect[0][3], rect[1][3]);
        rect[1][0] = 1;
        rect[1][1] = 0;
      }
      const size_t cell0 = (size_t)(li * 6 + face) * cells_per_face;
      for (int hh = 0; hh < 2; ++hh) {
        if (rect[hh][0] > rect[hh][1]) continue;
        const int nx = rect[hh][1] - rect[hh][0] + 1, n = nx * (rect[hh][3] - rect[hh][2] + 1);
        for (int k = lane; k < n; k += 64) {
          const size_t cell = cell0 + (size_t)(rect[hh][2] + k / nx) * Rr + rect[hh][0] + k % nx;
          const int slot = atomicAdd(&LL.cnt[cell], 1);
          if (slot < kLightListCap) LL.ids[cell * kLightListCap + slot] = j;
        }
      }
    }
  }
}

// (E_t): one thread per cell against the light's entries
__global__ void __launch_bounds__(256) k_bin_light_tri_escape(const RenderParams p) {
  const LightLists LL = p.lt;
  const int Rr = LL.R, cells_per_face = Rr * Rr;
  const int cell = blockIdx.x * 256 + (int)threadIdx.x;
  if (cell >= LL.n_listed * 6 * cells_per_face) return;
  const int li = cell / (6 * cells_per_face), face = (cell / cells_per_face) % 6;
  const int cw = (cell % cells_per_face) / Rr, cu = cell % Rr;
  int32_t *hdr0 = LL.hdr + (size_t)(li * 6) * kTileHdrInts;
  const int n_esc = hdr0[1];
  if (n_esc == 0) return;
  if (n_esc > kLightEscCap) {
    LL.hdr[(size_t)(li * 6 + face) * kTileHdrInts + 2] = 1; // too many: this light's rays take the sweep
    return;
  }
  int m, ia, ib;
  double sign;
  light_face_axes(face, m, ia, ib, sign);
  // directions of the cell: v = sign e_m + u e_a + w e_b, (u, w) in the cell's square grown a little
  const double u0 = 2.0 * (cu - 1e-3) / Rr - 1.0, u1 = 2.0 * (cu + 1.0 + 1e-3) / Rr - 1.0;
  const double w0 = 2.0 * (cw - 1e-3) / Rr - 1.0, w1 = 2.0 * (cw + 1.0 + 1e-3) / Rr - 1.0;
  const double um = fmax(fabs(u0), fabs(u1)), wm = fmax(fabs(w0), fabs(w1));
  const double vmax = sqrt(1.0 + um * um + wm * wm) * (1.0 + 1e-9);
  for (int k = 0; k < n_esc; ++k) {
    const LightEsc X = LL.esc[(size_t)li * kLightEscCap + k];
    const double nv[3] = {X.nx, X.ny, X.nz};
    const double fA = sign * nv[m], fU = nv[ia], fW = nv[ib];
    const double lo = fA + fmin(u0 * fU, u1 * fU) + fmin(w0 * fW, w1 * fW);
    const double hi = fA + fmax(u0 * fU, u1 * fU) + fmax(w0 * fW, w1 * fW);
    const double wd = (double)X.kp * vmax + 1e-12;
    if ((lo <= wd && hi >= -wd) || !(fA == fA)) {
      const int slot = atomicAdd(&LL.cnt[cell], 1);
      if (slot < kLightListCap) LL.ids[(size_t)cell * kLightListCap + slot] = X.pair;
    }
  }
}

// Shadow rays of a wave through the triangle light lists (the scheme of anyhit_sph_light_lists): four
// pair records per step through the triangle filter, flagged pairs through the reference arithmetic.
// `far` rays (outside the region the FILTER's margins hold for) make the wave take the sweep.
template <typename FetchF, typename FetchE>
DEVINL bool anyhit_tri_light_lists(const LightLists &LL, int cell, FetchF recf, FetchE rece, int base, f3 o, f3 L,
                                   const RayTF &rf, Any (&a)[1], int &n_tests, int &swept) {
  const V3<float> ov[1] = {{o.x, o.y, o.z}}, Lv[1] = {{L.x, L.y, L.z}};
  auto level2 = [&](int k0, int k1) { // pair records k0, k1 = sorted triangles 2 k0, 2 k0 + 1, 2 k1, 2 k1 + 1
    const TriPairF R[2] = {recf(k0), recf(k1)};
    v2f A[2], B[2], C[2];
    tripair2_any_filter_pk(R, rf, A, B, C);
    const int f0 = tri_flags(A[0], B[0], C[0], -1), f1 = tri_flags(A[1], B[1], C[1], -1);
    if (ANY_LANE_RARE(max(f0, f1) >= 0)) {
      if (__builtin_amdgcn_ballot_w64(f0 >= 0)) {
        test_tri_any<float, 1>(rece(2 * k0), base + 2 * k0, ov, Lv, a);
        test_tri_any<float, 1>(rece(2 * k0 + 1), base + 2 * k0 + 1, ov, Lv, a);
      }
      if (__builtin_amdgcn_ballot_w64(f1 >= 0)) {
        test_tri_any<float, 1>(rece(2 * k1), base + 2 * k1, ov, Lv, a);
        test_tri_any<float, 1>(rece(2 * k1 + 1), base + 2 * k1 + 1, ov, Lv, a);
      }
    }
  };
  const int cells_per_face = LL.R * LL.R;
  const ListPtr cnt = (ListPtr)(uintptr_t)LL.cnt;
  unsigned long long done = 0;
  for (;;) {
    const unsigned long long todo = __builtin_amdgcn_ballot_w64(a[0].tb > 0.f) & ~done;
    if (todo == 0) return true;
    const int leader = __builtin_ctzll(todo);
    const int c = __builtin_amdgcn_readlane(cell, leader);
    const ListPtr hdr = (ListPtr)(uintptr_t)(LL.hdr + (size_t)(c / cells_per_face) * kTileHdrInts);
    const int n_glob = hdr[0], n = cnt[c];
    if (n_glob > kTileGlobalCap || n > kLightListCap || hdr[2] != 0) return false;
    const SmemFetch<DevIdx4> glob{reinterpret_cast<const DevIdx4 *>(LL.hdr + (size_t)(c / cells_per_face) * kTileHdrInts + 8)};
    const SmemFetch<DevIdx4> ids{reinterpret_cast<const DevIdx4 *>(LL.ids + (size_t)c * kLightListCap)};
    auto batch = [&](const DevIdx4 &I) { // four pair records = 8 triangles
      n_tests += (a[0].tb > 0.f) ? 8 : 0;
      swept += 8;
      level2(I.v[0], I.v[1]);
      level2(I.v[2], I.v[3]);
    };
    for (int k = 0; k < n_glob; k += 4) {
      if (__builtin_amdgcn_ballot_w64(a[0].tb > 0.f) == 0) return true;
      batch(glob(k >> 2));
    }
    for (int k = 0; k < n; k += 4) {
      if (__builtin_amdgcn_ballot_w64(a[0].tb > 0.f) == 0) return true;
      batch(ids(k >> 2));
    }
    done |= __builtin_amdgcn_ballot_w64(cell == c);
  }
}

} // namespace esc
