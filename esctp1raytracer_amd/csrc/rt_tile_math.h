// rt_tile_math.h -- the geometry behind the tile lists (rt_lists.h), host + device: the camera frame
// in double, the pixel rectangle of a bounding sphere, the direction cone of a tile.  Shared with
// rt_capi.cpp, which exposes it to the CPU tests (esc_tile_rect / esc_tile_cone): what the tests
// check against per-pixel brute force is this very code.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rt_device.h"

namespace esc {

#define HDINL __host__ __device__ __forceinline__

struct CamD { // the camera frame in double: rows of [H V A]^-1 and what fp32 rounding can move a ray by
  double o[3];
  double b1[3], b2[3], b3[3];
  double nb1, nb2, nb3; // their lengths
  double eps_p;
  bool ok;
};
HDINL double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
HDINL void cross3(const double *a, const double *b, double *r) {
  r[0] = a[1] * b[2] - a[2] * b[1];
  r[1] = a[2] * b[0] - a[0] * b[2];
  r[2] = a[0] * b[1] - a[1] * b[0];
}
HDINL CamD cam_frame(const RenderParams &p) {
  CamD c;
  double Hh[3], V[3], A[3], m1 = 0.0;
  for (int k = 0; k < 3; ++k) {
    c.o[k] = p.origin[k];
    Hh[k] = p.horizontal[k];
    V[k] = p.vertical[k];
    A[k] = (double)p.llc[k] - c.o[k];
    m1 += fabs((double)p.llc[k]) + fabs(Hh[k]) + fabs(V[k]) + fabs(c.o[k]);
  }
  double hv[3], va[3], ah[3];
  cross3(Hh, V, hv);
  cross3(V, A, va);
  cross3(A, Hh, ah);
  const double det = dot3(A, hv);
  const double scale = sqrt(dot3(A, A) * dot3(Hh, Hh) * dot3(V, V));
  c.ok = fabs(det) > 1e-9 * scale && scale > 0.0 && scale < 1e150; // also false for NaN
  const double id = c.ok ? 1.0 / det : 0.0;
  for (int k = 0; k < 3; ++k) {
    c.b1[k] = va[k] * id;
    c.b2[k] = ah[k] * id;
    c.b3[k] = hv[k] * id;
  }
  c.nb1 = sqrt(dot3(c.b1, c.b1));
  c.nb2 = sqrt(dot3(c.b2, c.b2));
  c.nb3 = sqrt(dot3(c.b3, c.b3));
  c.eps_p = 0x1p-19 * m1;
  return c;
}

// image row of local row lr of the band (the Tile<> mapping of rt_kernels.hip)
HDINL int band_image_row(const RenderParams &p, int lr) {
  return p.h0 + (lr / p.strip_rows) * p.strip_step + (lr % p.strip_rows);
}
// local tile row (4 rows) that holds image rows [4 j, 4 j + 4), or -1; p.h0 is a multiple of 4
HDINL int band_tile_row(const RenderParams &p, int j) {
  const int rel = 4 * j - p.h0;
  if (rel < 0) return -1;
  int lr = rel;
  if (p.strip_step > 0) {
    const int k = rel / p.strip_step, off = rel % p.strip_step;
    if (off >= p.strip_rows) return -1;
    lr = k * p.strip_rows + off;
  }
  if (lr >= p.n_local_rows) return -1;
  return lr >> 2;
}

// Cube map around a light sample point P (rt_device.h LightLists).  Face f = 2 m + (negative ? 1 : 0),
// m the axis of the largest |component| of a direction v (lowest index on ties); its other two
// components over |v[m]|, in axis order, are the face coordinates (u, w) in [-1, 1]; cell = floor((u +
// 1) R / 2), floor((w + 1) R / 2).  As a "camera": rays p(s, t) = A + s H + t V with H = 2 e_a, V = 2 e_b,
// A = +-e_m - e_a - e_b, s = (u + 1) / 2, t = (w + 1) / 2, so that with W - 1 = H - 1 = R a "pixel
// coordinate" is a continuous cell coordinate.
HDINL void light_face_axes(int face, int &m, int &ia, int &ib, double &sign) {
  m = face >> 1;
  sign = (face & 1) ? -1.0 : 1.0;
  ia = (m == 0) ? 1 : 0;
  ib = (m == 2) ? 1 : 2;
}
HDINL CamD light_face_frame(const double P[3], int face) {
  int m, ia, ib;
  double sign;
  light_face_axes(face, m, ia, ib, sign);
  RenderParams q;
  for (int k = 0; k < 3; ++k) {
    q.origin[k] = 0.f;
    q.horizontal[k] = (k == ia) ? 2.f : 0.f;
    q.vertical[k] = (k == ib) ? 2.f : 0.f;
    q.llc[k] = (k == m) ? (float)sign : ((k == ia || k == ib) ? -1.f : 0.f);
  }
  CamD c = cam_frame(q);
  for (int k = 0; k < 3; ++k) c.o[k] = P[k];
  return c;
}

// Pixel extents ext = {wlo, whi, hlo, hhi} (double, NOT clipped to the image, grown by what fp32
// rounding can move a ray) outside which no primary ray's line passes within R of c (= C - o).
// false: unbounded (the camera plane cuts the sphere) or nothing can be said.  depth (optional)
// receives b_3 . c: its sign tells on which side of the camera plane the sphere lies (the lines of
// a sphere behind the camera cross the image point-mirrored -- a CONVEX HULL of several spheres'
// extents is only meaningful when all of them lie on one side).
HDINL bool sphere_pixel_extent(const CamD &cam, int W, int H, const double c[3], double R, double ext[4],
                               double *depth = nullptr, double pad_px = -1.0) {
  const double iR2 = 1.0 / (R * R);
  const double c1 = dot3(cam.b1, c), c2 = dot3(cam.b2, c), c3 = dot3(cam.b3, c);
  if (depth) *depth = c3;
  const double q33 = cam.nb3 * cam.nb3 - c3 * c3 * iR2;
  if (!(q33 < -1e-9 * cam.nb3 * cam.nb3)) return false; // the camera plane cuts the sphere (or NaN)
  const double q11 = cam.nb1 * cam.nb1 - c1 * c1 * iR2, q13 = dot3(cam.b1, cam.b3) - c1 * c3 * iR2;
  const double q22 = cam.nb2 * cam.nb2 - c2 * c2 * iR2, q23 = dot3(cam.b2, cam.b3) - c2 * c3 * iR2;
  const double ds = sqrt(fmax(0.0, q13 * q13 - q11 * q33)), dt = sqrt(fmax(0.0, q23 * q23 - q22 * q33));
  const double sa = (q13 - ds) / q33, sb = (q13 + ds) / q33;
  const double ta = (q23 - dt) / q33, tb = (q23 + dt) / q33;
  const double W1 = (double)(W - 1), H1 = (double)(H - 1);
  // pad_px >= 0: the caller's own pad (light-space maps: the lookup is the library's, not the
  // reference's pixel grid)
  const double pw = pad_px >= 0.0 ? pad_px : 1.0 + (cam.nb1 + cam.nb3) * cam.eps_p * 1.01 * W1;
  const double ph = pad_px >= 0.0 ? pad_px : 1.0 + (cam.nb2 + cam.nb3) * cam.eps_p * 1.01 * H1;
  const double big = 1e9;
  ext[0] = fmax(-big, fmin(big, fmin(sa, sb) * W1 - pw));
  ext[1] = fmax(-big, fmin(big, fmax(sa, sb) * W1 + pw));
  ext[2] = fmax(-big, fmin(big, fmin(ta, tb) * H1 - ph));
  ext[3] = fmax(-big, fmin(big, fmax(ta, tb) * H1 + ph));
  return ext[0] == ext[0] && ext[1] == ext[1] && ext[2] == ext[2] && ext[3] == ext[3]; // NaN: no
}
// extents -> the pixel rectangle clipped to the image; 1: rectangle, 2: wholly off screen
HDINL int extent_rect(const double ext[4], int W, int H, int &w0, int &w1, int &h0, int &h1) {
  w0 = (int)floor(ext[0]) < 0 ? 0 : (int)floor(ext[0]);
  w1 = (int)ceil(ext[1]) > W - 1 ? W - 1 : (int)ceil(ext[1]);
  h0 = (int)floor(ext[2]) < 0 ? 0 : (int)floor(ext[2]);
  h1 = (int)ceil(ext[3]) > H - 1 ? H - 1 : (int)ceil(ext[3]);
  return (w0 > w1 || h0 > h1) ? 2 : 1;
}
// both: 0 unbounded / unusable, 1 rectangle, 2 wholly off screen
HDINL int sphere_pixel_rect(const CamD &cam, int W, int H, const double c[3], double R, int &w0, int &w1,
                            int &h0, int &h1) {
  double ext[4];
  if (!sphere_pixel_extent(cam, W, H, c, R, ext)) return 0;
  return extent_rect(ext, W, H, w0, w1, h0, h1);
}

// The image-plane rectangle [s0, s1] x [t0, t1] that holds the (s, t) of every ray the reference
// computes for the pixels [32 tx, 32 tx + 32) x [h, h + 4) (grown like the sphere rectangles), and
// pmax >= |A + s H + t V| over it (|p| is convex: the largest value is at a corner).
HDINL void tile_st(const RenderParams &p, const CamD &cam, int tx, int h, double st[4]) {
  const double W1 = (double)(p.W - 1), H1 = (double)(p.H - 1);
  const double pw = 1.0 + (cam.nb1 + cam.nb3) * cam.eps_p * 1.01 * W1;
  const double ph = 1.0 + (cam.nb2 + cam.nb3) * cam.eps_p * 1.01 * H1;
  st[0] = (32.0 * tx - pw) / W1;
  st[1] = (32.0 * tx + 31.0 + pw) / W1;
  st[2] = ((double)h - ph) / H1;
  st[3] = ((double)h + 3.0 + ph) / H1;
}
HDINL double tile_pmax(const RenderParams &p, const CamD &cam, const double st[4]) {
  double pmax = 0.0;
  for (int k = 0; k < 4; ++k) {
    double q[3];
    for (int j = 0; j < 3; ++j)
      q[j] = ((double)p.llc[j] - cam.o[j]) + ((k & 1) ? st[1] : st[0]) * (double)p.horizontal[j] +
             ((k & 2) ? st[3] : st[2]) * (double)p.vertical[j];
    pmax = fmax(pmax, sqrt(dot3(q, q)));
  }
  return pmax * (1.0 + 1e-9);
}
HDINL void tile_st_rect(const RenderParams &p, const CamD &cam, int tx, int h, double st[4], double &pmax) {
  tile_st(p, cam, tx, h, st);
  pmax = tile_pmax(p, cam, st);
}
// can a ray of that rectangle have |d . n| <= kp, d = p / |p|?  p . n = fA + s fH + t fV is affine in
// (s, t), so its range over the rectangle is spanned by the corners
HDINL bool tile_band_hit(const double st[4], double pmax, double fA, double fH, double fV, double kp) {
  const double lo = fA + fmin(st[0] * fH, st[1] * fH) + fmin(st[2] * fV, st[3] * fV);
  const double hi = fA + fmax(st[0] * fH, st[1] * fH) + fmax(st[2] * fV, st[3] * fV);
  const double w = kp * pmax + 1e-12 * (fabs(fA) + fabs(fH) + fabs(fV));
  return lo <= w && hi >= -w; // false for NaN: the caller must have ruled those out
}

// The tiles [tx0, tx1] of the tile row that holds image rows [h, h + 4) which a ray with |p . n| <= kp |p|
// can fall into, p . n = fA + s fH + t fV (affine in the image-plane coordinates): over the row's t range
// the band is an interval of s.  Returns 1 (tx0 <= tx1: those tiles; tx0 > tx1: none) or 0 (nothing can
// be said: the caller switches the lists off).
// (pmax_row: the larger of the two end tiles' pmax -- |p| is convex, over the row strip it peaks at an
// end; t0, t1: the row's t range, tile_st's st[2], st[3])
HDINL int band_row_tiles_with(const RenderParams &p, const CamD &cam, int tiles_x, double pmax, double t0,
                              double t1, double fA, double fH, double fV, double kp, int &tx0, int &tx1) {
  const double w = kp * pmax + 1e-12 * (fabs(fA) + fabs(fH) + fabs(fV));
  const double g_lo = fA + fmin(t0 * fV, t1 * fV), g_hi = fA + fmax(t0 * fV, t1 * fV);
  if (!(pmax == pmax) || !(pmax < 1e150) || !(w == w) || !(g_lo == g_lo) || !(g_hi == g_hi)) return 0;
  // some t in the row with |g(t) + s fH| <= w  <=>  s fH in [-w - g_hi, w - g_lo]
  const double lo = -w - g_hi, hi = w - g_lo;
  const double W1 = (double)(p.W - 1);
  const double pw = 1.0 + (cam.nb1 + cam.nb3) * cam.eps_p * 1.01 * W1; // (tile_st's growth)
  tx0 = 0;
  tx1 = tiles_x - 1;
  if (fabs(fH) * (W1 + 2.0 * pw) > 1e-280) {
    const double sa = lo / fH, sb = hi / fH;
    const double wa = fmin(sa, sb) * W1 - pw - 31.0, wb = fmax(sa, sb) * W1 + pw; // tile tx holds [32 tx, 32 tx + 31]
    if (!(wa == wa) || !(wb == wb)) return 0;
    tx0 = (int)fmax(0.0, fmin(1e9, ceil(wa / 32.0))); // 32 tx + 31 >= the band's first pixel
    tx1 = (int)fmin((double)(tiles_x - 1), fmax(-1.0, floor(wb / 32.0)));
  } else if (!(lo <= 0.0 && hi >= 0.0)) {
    tx0 = 1; // the band is (all but) horizontal and misses this row
    tx1 = 0;
  }
  return 1;
}
HDINL int band_row_tiles(const RenderParams &p, const CamD &cam, int h, int tiles_x, double fA, double fH,
                         double fV, double kp, int &tx0, int &tx1) {
  double st[4], pmax_l, pmax_r;
  tile_st_rect(p, cam, 0, h, st, pmax_l);
  const double t0 = st[2], t1 = st[3];
  tile_st_rect(p, cam, tiles_x - 1, h, st, pmax_r);
  return band_row_tiles_with(p, cam, tiles_x, fmax(pmax_l, pmax_r), t0, t1, fA, fH, fV, kp, tx0, tx1);
}

} // namespace esc
