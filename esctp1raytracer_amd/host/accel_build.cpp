// accel_build.cpp -- binned-SAH BVH over padded primitive boxes (see accel_build.h).
//
// Why the boxes are padded.  The kernels must return exactly what the brute-force loops return,
// and those accept a primitive on the strength of ROUNDED fp32 arithmetic: a ray that misses the
// geometric primitive by a hair can still be a hit.  A box test may therefore only cull a
// primitive when the ray stays further away from it than that arithmetic can be wrong.
// With u = 2^-24 (fp32 unit roundoff) and M = the largest distance between the primitive and any
// point a ray can start from (OriginBounds):
//
//  * sphere (oc = o - c, b = dot(oc,d), disc = b*b - (dot(oc,oc) - r*r)):
//    three roundings per dot product, one per subtraction and product, |d|^2 within 6u of 1:
//        |disc - (r^2 - dist^2)| <= 18 u |oc|^2 + 2 u r^2,     dist = distance centre <-> ray line
//    so disc >= 0 implies dist^2 <= r^2 + 18 u M^2 + ..., and the accepted t2 = -b -+ sqrt(disc)
//    puts o + t2 d within sqrt(r^2 + 28 u M^2) of the centre.  The box is the cube around the
//    centre with half-width  R = sqrt(r^2 + 32 u M^2) + 16 u M  (the linear term also absorbs the
//    <= 2 u M by which a shadow ray's rounded direction leaves the exact line to its light point,
//    which the light-space bins are laid out on).
//
//  * triangle (ray_triangle.h:7-57, Cramer's rule, numerators in fp32): the numerators
//    dot(tvec,pvec), dot(dir,qvec) carry an absolute error of ~3 u |tvec| |edge| that does not
//    shrink with det, so the accepted (u,v) can sit  ~6 u M / sin(theta)  outside the triangle,
//    theta = angle between ray and triangle plane.  No finite pad covers theta -> 0; the pad used,
//    2^-12 M, covers every ray steeper than ~1.5e-3 rad by the worst-case bound (about ten times
//    shallower with typical rounding).  Rays that graze a triangle's plane more closely than that
//    get rounding noise for (u,v,t) from the reference arithmetic itself; for those -- and only
//    those -- the BVH mode may in principle cull a hit that brute force reports.  The parity tests
//    compare whole frames of both modes (tests/test_gpu_parity.py), DESIGN.md states the caveat.
//
// The slab test itself (rt_accel.h slab()) is off by < 4 u M in position, far inside either
// pad.  All pads are computed in double and rounded outward.
#include "accel_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <atomic>
#include <numeric>
#include <thread>

namespace esc {

namespace {

constexpr double kU = 5.9604644775390625e-08; // 2^-24

float down(double x) {
  float f = (float)x;
  return ((double)f > x) ? std::nextafterf(f, -FLT_MAX) : f;
}
float up(double x) {
  float f = (float)x;
  return ((double)f < x) ? std::nextafterf(f, FLT_MAX) : f;
}

// largest distance from `c` to a corner of the origin bounds
double far_corner(const OriginBounds &ob, const double c[3]) {
  double s = 0;
  for (int a = 0; a < 3; a++) {
    const double d = std::max(std::fabs(c[a] - ob.lo[a]), std::fabs(c[a] - ob.hi[a]));
    s += d * d;
  }
  return std::sqrt(s);
}

} // namespace

OriginBounds origin_bounds(const std::vector<DevTri> &tri, const std::vector<DevSph> &sph,
                           const std::vector<float> &light_points_xyz0, const float origin[3]) {
  OriginBounds ob;
  for (int a = 0; a < 3; a++) ob.lo[a] = ob.hi[a] = origin[a];
  auto grow = [&](double x, double y, double z, double r) {
    const double p[3] = {x, y, z};
    for (int a = 0; a < 3; a++) {
      ob.lo[a] = std::min(ob.lo[a], p[a] - r);
      ob.hi[a] = std::max(ob.hi[a], p[a] + r);
    }
  };
  for (const DevTri &t : tri) {
    grow(t.v0[0], t.v0[1], t.v0[2], 0);
    grow((double)t.v0[0] + t.e1[0], (double)t.v0[1] + t.e1[1], (double)t.v0[2] + t.e1[2], 0);
    grow((double)t.v0[0] + t.e2[0], (double)t.v0[1] + t.e2[1], (double)t.v0[2] + t.e2[2], 0);
  }
  for (const DevSph &s : sph) grow(s.cx, s.cy, s.cz, std::sqrt(std::max(0.0, (double)s.r2)));
  for (size_t i = 0; i + 4 <= light_points_xyz0.size(); i += 4)
    grow(light_points_xyz0[i], light_points_xyz0[i + 1], light_points_xyz0[i + 2], 0);
  double diag = 0;
  for (int a = 0; a < 3; a++) diag += (ob.hi[a] - ob.lo[a]) * (ob.hi[a] - ob.lo[a]);
  diag = std::sqrt(diag);
  if (light_points_xyz0.size() > 4) {
    // More than one light point: quirk S3 starts the NEXT light's shadow ray at
    // camera + dir * (t_occ - eps), t_occ = the previous light's occluder distance along ITS shadow
    // ray (main.cpp:757 with the t occlusion() left behind) -- a point on the primary ray that can
    // lie past the hit surface and outside the scene box.  Light 2's origin o_2 has t_occ < light 1's
    // shadow-ray length <= the diagonal of (scene + lights + camera): within `diag` of the camera.
    // Light k's ray can be as long as |P_(k-1) - cam| + |cam - o_(k-1)|, so o_k lies within (k - 1) diag
    // of the camera: the ball takes the number of light POINTS (>= lights) less one.
    ob.ball_mult = (double)(light_points_xyz0.size() / 4 - 1);
    ob.ball = ob.ball_mult * diag;
    for (int a = 0; a < 3; a++) ob.cam[a] = origin[a];
    for (int a = 0; a < 3; a++) {
      ob.lo[a] = std::min(ob.lo[a], (double)origin[a] - ob.ball);
      ob.hi[a] = std::max(ob.hi[a], (double)origin[a] + ob.ball);
    }
    diag = 0;
    for (int a = 0; a < 3; a++) diag += (ob.hi[a] - ob.lo[a]) * (ob.hi[a] - ob.lo[a]);
    diag = std::sqrt(diag);
  }
  const double g = 0.1 * diag + 1e-6;
  for (int a = 0; a < 3; a++) {
    ob.lo[a] -= g;
    ob.hi[a] += g;
  }
  return ob;
}

void triangle_boxes(const std::vector<DevTri> &tri, const OriginBounds &ob,
                    std::vector<PrimBox> &out) {
  out.resize(tri.size());
  for (size_t i = 0; i < tri.size(); i++) {
    const DevTri &t = tri[i];
    double lo[3], hi[3], c[3];
    for (int a = 0; a < 3; a++) {
      const double p0 = t.v0[a], p1 = (double)t.v0[a] + t.e1[a], p2 = (double)t.v0[a] + t.e2[a];
      lo[a] = std::min(p0, std::min(p1, p2));
      hi[a] = std::max(p0, std::max(p1, p2));
      c[a] = 0.5 * (lo[a] + hi[a]);
    }
    double ext = 0;
    for (int a = 0; a < 3; a++) ext += (hi[a] - lo[a]) * (hi[a] - lo[a]);
    const double M = far_corner(ob, c) + 0.5 * std::sqrt(ext);
    const double pad = M * (1.0 / 4096.0);
    for (int a = 0; a < 3; a++) {
      out[i].lo[a] = down(lo[a] - pad);
      out[i].hi[a] = up(hi[a] + pad);
    }
  }
}

void sphere_boxes(const std::vector<DevSph> &sph, const OriginBounds &ob,
                  std::vector<PrimBox> &out) {
  out.resize(sph.size());
  for (size_t i = 0; i < sph.size(); i++) {
    const DevSph &s = sph[i];
    const double c[3] = {s.cx, s.cy, s.cz};
    const double r2 = std::max(0.0, (double)s.r2);
    const double M = far_corner(ob, c) + std::sqrt(r2);
    const double R = std::sqrt(r2 * (1.0 + 4 * kU) + 32.0 * kU * M * M) + 16.0 * kU * M;
    for (int a = 0; a < 3; a++) {
      out[i].lo[a] = down(c[a] - R);
      out[i].hi[a] = up(c[a] + R);
    }
  }
}

namespace {

struct Sub { // what a finished subtree reports to its parent
  int32_t code;
  float lo[3], hi[3];
  uint32_t minkey;
  int depth; // internal nodes on its longest path
};

struct Builder {
  const std::vector<PrimBox> &boxes;
  const int block;
  const uint32_t key_base;
  const int max_depth;
  BuiltBvh &out;
  std::vector<int32_t> &idx;     // shared permutation; a builder only touches its own range
  const std::vector<float> &cen; // 3 per primitive

  // internal levels a count-balanced (median) subtree over n primitives needs
  int levels_needed(size_t n) const {
    int l = 0;
    while (n > (size_t)block) {
      n = (n + 1) / 2;
      l++;
    }
    return l;
  }

  static double area(const float lo[3], const float hi[3]) {
    const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  }

  Sub leaf(size_t b, size_t e) {
    Sub s;
    s.code = ~out.n_blocks;
    s.depth = 0;
    s.minkey = 0xFFFFFFFFu;
    for (int a = 0; a < 3; a++) {
      s.lo[a] = FLT_MAX;
      s.hi[a] = -FLT_MAX;
    }
    for (size_t i = b; i < e; i++) {
      const int32_t k = idx[i];
      out.order.push_back(k);
      s.minkey = std::min(s.minkey, key_base + (uint32_t)k);
      for (int a = 0; a < 3; a++) {
        s.lo[a] = std::min(s.lo[a], boxes[k].lo[a]);
        s.hi[a] = std::max(s.hi[a], boxes[k].hi[a]);
      }
    }
    for (size_t i = e - b; i < (size_t)block; i++) out.order.push_back(-1);
    out.n_blocks++;
    return s;
  }

  // returns the split position m (b < m < e) after partitioning idx[b,e)
  size_t split(size_t b, size_t e, int depth_used) {
    const size_t n = e - b;
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = b; i < e; i++)
      for (int a = 0; a < 3; a++) {
        const float c = cen[3 * (size_t)idx[i] + a];
        clo[a] = std::min(clo[a], c);
        chi[a] = std::max(chi[a], c);
      }
    int wide = 0;
    for (int a = 1; a < 3; a++)
      if (chi[a] - clo[a] > chi[wide] - clo[wide]) wide = a;

    auto median = [&]() {
      const size_t m = b + (n + 1) / 2;
      std::nth_element(idx.begin() + b, idx.begin() + m, idx.begin() + e,
                       [&](int32_t x, int32_t y) {
                         const float cx = cen[3 * (size_t)x + wide], cy = cen[3 * (size_t)y + wide];
                         return cx < cy || (cx == cy && x < y);
                       });
      return m;
    };
    // depth budget: a median subtree finishes in levels_needed(n) levels; SAH may be lopsided
    if (depth_used + levels_needed(n) >= max_depth || !(chi[wide] > clo[wide])) return median();

    constexpr int NB = 16;
    struct Bin {
      float lo[3], hi[3];
      size_t n;
    };
    Bin bins[3][NB]; // all three axes binned in ONE pass over the primitives
    for (auto &ax : bins)
      for (auto &bn : ax) {
        bn.n = 0;
        for (int k = 0; k < 3; k++) {
          bn.lo[k] = FLT_MAX;
          bn.hi[k] = -FLT_MAX;
        }
      }
    double scale3[3];
    for (int a = 0; a < 3; a++)
      scale3[a] = (chi[a] > clo[a]) ? NB / ((double)chi[a] - clo[a]) : 0.0;
    for (size_t i = b; i < e; i++) {
      const int32_t k = idx[i];
      const PrimBox &pb = boxes[(size_t)k];
      for (int a = 0; a < 3; a++) {
        int bi = (int)(((double)cen[3 * (size_t)k + a] - clo[a]) * scale3[a]);
        bi = std::min(std::max(bi, 0), NB - 1);
        Bin &bn = bins[a][bi];
        bn.n++;
        for (int q = 0; q < 3; q++) {
          bn.lo[q] = std::min(bn.lo[q], pb.lo[q]);
          bn.hi[q] = std::max(bn.hi[q], pb.hi[q]);
        }
      }
    }
    double best = DBL_MAX;
    int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; a++) {
      if (!(chi[a] > clo[a])) continue;
      const Bin *bn = bins[a];
      double right_area[NB];
      size_t right_n[NB];
      {
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        size_t cnt = 0;
        for (int j = NB - 1; j >= 1; j--) {
          if (bn[j].n) {
            for (int q = 0; q < 3; q++) {
              lo[q] = std::min(lo[q], bn[j].lo[q]);
              hi[q] = std::max(hi[q], bn[j].hi[q]);
            }
            cnt += bn[j].n;
          }
          right_area[j] = cnt ? area(lo, hi) : 0.0;
          right_n[j] = cnt;
        }
      }
      float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      size_t cnt = 0;
      for (int j = 0; j + 1 < NB; j++) { // split between bin j and j+1
        if (bn[j].n) {
          for (int q = 0; q < 3; q++) {
            lo[q] = std::min(lo[q], bn[j].lo[q]);
            hi[q] = std::max(hi[q], bn[j].hi[q]);
          }
          cnt += bn[j].n;
        }
        const size_t nr = right_n[j + 1];
        if (cnt == 0 || nr == 0) continue;
        const double cost = area(lo, hi) * (double)((cnt + block - 1) / block) +
                            right_area[j + 1] * (double)((nr + block - 1) / block);
        if (cost < best) {
          best = cost;
          best_axis = a;
          best_bin = j;
        }
      }
    }
    if (best_axis < 0) return median();
    const double scale = NB / ((double)chi[best_axis] - clo[best_axis]);
    auto mid = std::partition(idx.begin() + b, idx.begin() + e, [&](int32_t k) {
      int bi = (int)(((double)cen[3 * (size_t)k + best_axis] - clo[best_axis]) * scale);
      bi = std::min(std::max(bi, 0), NB - 1);
      return bi <= best_bin;
    });
    const size_t m = (size_t)(mid - idx.begin());
    if (m == b || m == e) return median();
    const size_t big = std::max(m - b, e - m);
    if (depth_used + 1 + levels_needed(big) > max_depth) return median();
    return m;
  }

  Sub build(size_t b, size_t e, int depth_used) {
    if (e - b <= (size_t)block) return leaf(b, e);
    const int32_t me = (int32_t)out.nodes.size();
    out.nodes.emplace_back();
    const size_t m = split(b, e, depth_used);
    const Sub L = build(b, m, depth_used + 1);
    const Sub R = build(m, e, depth_used + 1);
    BvhNode &N = out.nodes[(size_t)me];
    std::memcpy(N.lo0, L.lo, 12);
    std::memcpy(N.hi0, L.hi, 12);
    std::memcpy(N.lo1, R.lo, 12);
    std::memcpy(N.hi1, R.hi, 12);
    N.child[0] = L.code;
    N.child[1] = R.code;
    N.minkey[0] = L.minkey;
    N.minkey[1] = R.minkey;
    Sub s;
    s.code = me;
    s.depth = 1 + std::max(L.depth, R.depth);
    s.minkey = std::min(L.minkey, R.minkey);
    for (int a = 0; a < 3; a++) {
      s.lo[a] = std::min(L.lo[a], R.lo[a]);
      s.hi[a] = std::max(L.hi[a], R.hi[a]);
    }
    return s;
  }
};

} // namespace

namespace {

// top of a parallel build: the first few splits are made serially, every range below them is
// a task some thread builds into its own BuiltBvh; the pieces are then renumbered into one tree
struct TopNode {
  int child[2]; // >= 0: TopNode index; < 0: ~task
};
struct Task {
  size_t b, e;
  int depth;
  BuiltBvh sub;
  Sub result;
};

} // namespace

void build_bvh(const std::vector<PrimBox> &boxes, int block, uint32_t key_base, int max_depth,
               BuiltBvh &out) {
  out.nodes.clear();
  out.order.clear();
  out.root = 0;
  out.depth = 0;
  out.n_blocks = 0;
  if (boxes.empty()) return;
  std::vector<int32_t> idx(boxes.size());
  std::iota(idx.begin(), idx.end(), 0);
  std::vector<float> cen(3 * boxes.size());
  for (size_t i = 0; i < boxes.size(); i++)
    for (int a = 0; a < 3; a++) cen[3 * i + a] = 0.5f * boxes[i].lo[a] + 0.5f * boxes[i].hi[a];

  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (boxes.size() < 16384 || hw < 2) { // small sets: one thread, pre-order numbering
    Builder B{boxes, block, key_base, max_depth, out, idx, cen};
    out.nodes.reserve(2 * boxes.size() / (size_t)block + 2);
    const Sub r = B.build(0, boxes.size(), 0);
    out.root = r.code;
    out.depth = r.depth;
    return;
  }

  // ---- serial top: split until the ranges are small enough to hand out
  std::vector<TopNode> top;
  std::vector<Task> tasks;
  const size_t grain = std::max<size_t>(boxes.size() / (2 * hw), 4096);
  BuiltBvh scratch; // the top splits only permute idx; they emit nothing
  Builder T{boxes, block, key_base, max_depth, scratch, idx, cen};
  struct Local {
    static int make(Builder &T, std::vector<TopNode> &top, std::vector<Task> &tasks, size_t b,
                    size_t e, int depth, size_t grain) {
      if (e - b <= grain || depth >= 8) {
        tasks.push_back(Task{b, e, depth, {}, {}});
        return ~(int)(tasks.size() - 1);
      }
      const int me = (int)top.size();
      top.push_back(TopNode{{0, 0}});
      const size_t m = T.split(b, e, depth);
      const int l = make(T, top, tasks, b, m, depth + 1, grain);
      const int r = make(T, top, tasks, m, e, depth + 1, grain);
      top[(size_t)me].child[0] = l;
      top[(size_t)me].child[1] = r;
      return me;
    }
  };
  const int top_root = Local::make(T, top, tasks, 0, boxes.size(), 0, grain);

  // ---- the tasks, in parallel (disjoint ranges of idx, private outputs)
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= tasks.size()) return;
      Task &t = tasks[k];
      Builder B{boxes, block, key_base, max_depth, t.sub, idx, cen};
      t.result = B.build(t.b, t.e, t.depth);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned i = 1; i < hw; i++) pool.emplace_back(worker);
  worker();
  for (auto &th : pool) th.join();

  // ---- stitch: top nodes first (creation order), then each task's nodes and blocks in task order
  std::vector<int32_t> node_off(tasks.size()), block_off(tasks.size());
  int32_t n_nodes = (int32_t)top.size(), n_blocks = 0;
  for (size_t k = 0; k < tasks.size(); k++) {
    node_off[k] = n_nodes;
    block_off[k] = n_blocks;
    n_nodes += (int32_t)tasks[k].sub.nodes.size();
    n_blocks += tasks[k].sub.n_blocks;
  }
  out.nodes.resize((size_t)n_nodes);
  out.order.reserve((size_t)n_blocks * (size_t)block);
  out.n_blocks = n_blocks;
  auto fix = [&](int32_t code, size_t k) {
    return code >= 0 ? code + node_off[k] : ~((~code) + block_off[k]);
  };
  for (size_t k = 0; k < tasks.size(); k++) {
    const BuiltBvh &sb = tasks[k].sub;
    for (size_t i = 0; i < sb.nodes.size(); i++) {
      BvhNode n = sb.nodes[i];
      n.child[0] = fix(n.child[0], k);
      n.child[1] = fix(n.child[1], k);
      out.nodes[(size_t)node_off[k] + i] = n;
    }
    out.order.insert(out.order.end(), sb.order.begin(), sb.order.end());
    tasks[k].result.code = fix(tasks[k].result.code, k);
  }
  struct Stitch {
    static Sub go(const std::vector<TopNode> &top, const std::vector<Task> &tasks, BuiltBvh &out,
                  int code) {
      if (code < 0) return tasks[(size_t)~code].result;
      const Sub L = go(top, tasks, out, top[(size_t)code].child[0]);
      const Sub R = go(top, tasks, out, top[(size_t)code].child[1]);
      BvhNode &N = out.nodes[(size_t)code];
      std::memcpy(N.lo0, L.lo, 12);
      std::memcpy(N.hi0, L.hi, 12);
      std::memcpy(N.lo1, R.lo, 12);
      std::memcpy(N.hi1, R.hi, 12);
      N.child[0] = L.code;
      N.child[1] = R.code;
      N.minkey[0] = L.minkey;
      N.minkey[1] = R.minkey;
      Sub s;
      s.code = code;
      s.depth = 1 + std::max(L.depth, R.depth);
      s.minkey = std::min(L.minkey, R.minkey);
      for (int a = 0; a < 3; a++) {
        s.lo[a] = std::min(L.lo[a], R.lo[a]);
        s.hi[a] = std::max(L.hi[a], R.hi[a]);
      }
      return s;
    }
  };
  const Sub r = Stitch::go(top, tasks, out, top_root);
  out.root = r.code;
  out.depth = r.depth;
}

// ---- sphere groups of the brute-force primary pass (rt_device.h SphGroups) ----------------
// Spatial order by k-d median splits along the longest axis of the centres.  Every left part is a
// whole number of `huge` runs while more than one is left, then of `big` runs, then of `run`s, so
// consecutive runs of `huge` (hyper-groups), `big` (super-groups) and `run` (groups) in the result
// are subtrees.  Ties are broken by
// index: the order is a function of the scene alone.
void group_order_points(const std::vector<float> &xyz, int run, int big, int huge,
                        std::vector<int32_t> &order) {
  const size_t n_pts = xyz.size() / 3;
  order.resize(n_pts);
  std::iota(order.begin(), order.end(), 0);
  struct Split {
    static void go(const float *xyz, int small, int big, int huge, int32_t *idx, size_t n) {
      const int run = n > (size_t)huge ? huge : (n > (size_t)big ? big : small);
      if (n <= (size_t)small) {
        std::sort(idx, idx + n);
        return;
      }
      float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (size_t i = 0; i < n; i++) {
        const float *c = xyz + 3 * (size_t)idx[i];
        for (int a = 0; a < 3; a++) {
          lo[a] = std::min(lo[a], c[a]);
          hi[a] = std::max(hi[a], c[a]);
        }
      }
      int ax = 0;
      for (int a = 1; a < 3; a++)
        if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
      const size_t runs = (n + (size_t)run - 1) / (size_t)run;
      const size_t left = (runs / 2) * (size_t)run; // 0 < left < n since runs >= 2
      // a TOTAL order also for non-finite coordinates (a NaN vertex gives a NaN centroid; `<` on
      // NaNs is not a strict weak ordering, which nth_element requires): keys are the fp32 bit
      // patterns mapped monotonically to unsigned integers, NaNs after +inf; ties by index
      auto key = [&](int32_t i) {
        uint32_t u;
        const float c = xyz[3 * (size_t)i + ax];
        std::memcpy(&u, &c, 4);
        if (c != c) return 0xffffffffu;                      // NaN: last
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // -x ... -0 +0 ... +x, ascending
      };
      std::nth_element(idx, idx + left, idx + n, [&](int32_t a, int32_t b) {
        const uint32_t ka = key(a), kb = key(b);
        return ka < kb || (ka == kb && a < b);
      });
      go(xyz, small, big, huge, idx, left);
      go(xyz, small, big, huge, idx + left, n - left);
    }
  };
  Split::go(xyz.data(), run, big, huge, order.data(), order.size());
}

void group_order(const std::vector<DevSph> &sph, int run, int big, int huge,
                 std::vector<int32_t> &order) {
  std::vector<float> xyz(3 * sph.size());
  for (size_t i = 0; i < sph.size(); i++) {
    xyz[3 * i + 0] = sph[i].cx;
    xyz[3 * i + 1] = sph[i].cy;
    xyz[3 * i + 2] = sph[i].cz;
  }
  group_order_points(xyz, run, big, huge, order);
}

void group_order(const std::vector<DevTri> &tri, int run, int big, int huge,
                 std::vector<int32_t> &order) {
  std::vector<float> xyz(3 * tri.size());
  for (size_t i = 0; i < tri.size(); i++)
    for (int a = 0; a < 3; a++) // the centroid (non-finite ones sort last: group_order_points' keys)
      xyz[3 * i + a] = tri[i].v0[a] + (tri[i].e1[a] + tri[i].e2[a]) * (1.f / 3.f);
  group_order_points(xyz, run, big, huge, order);
}

// Bounding sphere of the spheres order[first .. first + count): centre = middle of the box around
// them (stored in fp32), rgeo >= r_i + |c_i - C| for every member, rounded up.
DevSphGroup group_bounds(const std::vector<DevSph> &sph, const int32_t *order, int count) {
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (int i = 0; i < count; i++) {
    const DevSph &s = sph[(size_t)order[i]];
    const double r = std::sqrt(std::max((double)s.r2, 0.0));
    const double c[3] = {s.cx, s.cy, s.cz};
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], c[a] - r);
      hi[a] = std::max(hi[a], c[a] + r);
    }
  }
  DevSphGroup G;
  G.cx = (float)(0.5 * (lo[0] + hi[0]));
  G.cy = (float)(0.5 * (lo[1] + hi[1]));
  G.cz = (float)(0.5 * (lo[2] + hi[2]));
  double rg = 0.0;
  for (int i = 0; i < count; i++) {
    const DevSph &s = sph[(size_t)order[i]];
    const double dx = (double)s.cx - G.cx, dy = (double)s.cy - G.cy, dz = (double)s.cz - G.cz;
    rg = std::max(rg, std::sqrt(std::max((double)s.r2, 0.0)) + std::sqrt(dx * dx + dy * dy + dz * dz));
  }
  rg *= 1.0 + 0x1p-40;
  float rf = (float)rg;
  if ((double)rf < rg) rf = std::nextafterf(rf, HUGE_VALF);
  G.rgeo = rf; // NaN / inf in, NaN / inf out: the group is then always a candidate
  return G;
}

// Static record of the triangles order[0 .. count) as one group (rt_device.h DevTriGroup), in
// double, every bound rounded up.  The node's slack factor k in [1, slack_cap] trades the two
// halves of the pre-filter against each other (rt_brute.h "Triangle GROUPS", (K)): the escape
// threshold is tau_t / k, and rgeo >= k rho_t + max_v |v - C| over the members' vertices.  Per member: centroid G, bounding radius rho around it, longest
// edge emax, n1 = e2 x e1; a member with rho <= 2^-9.9 emax (a sliver: the pre-filter passes those
// on unconditionally) or without a normal makes the group `always` open.
DevTriGroup tri_group_bounds(const std::vector<DevTri> &tri, const int32_t *order, int count,
                             double slack_cap) {
  const double u = 0x1p-24;
  struct M {
    double G[3], rho, nh[3], b0, b1, ext[3], a12, vtx[3][3];
    bool bad;
  };
  std::vector<M> ms((size_t)count);
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  bool always = false;
  for (int i = 0; i < count; i++) {
    const DevTri &t = tri[(size_t)order[i]];
    M &m = ms[(size_t)i];
    double e1[3], e2[3], s3[3], r0 = 0, r1 = 0, r2 = 0, l1 = 0, l2 = 0, a1 = 0, a2 = 0;
    for (int a = 0; a < 3; a++) {
      e1[a] = t.e1[a];
      e2[a] = t.e2[a];
      s3[a] = (e1[a] + e2[a]) / 3.0;
      m.G[a] = (double)t.v0[a] + s3[a];
      m.ext[a] = t.v0[a];
      r0 += s3[a] * s3[a];
      r1 += (e1[a] - s3[a]) * (e1[a] - s3[a]);
      r2 += (e2[a] - s3[a]) * (e2[a] - s3[a]);
      l1 += e1[a] * e1[a];
      l2 += e2[a] * e2[a];
      a1 += std::fabs(e1[a]);
      a2 += std::fabs(e2[a]);
    }
    m.rho = std::sqrt(std::max(r0, std::max(r1, r2))) * 1.00002;
    m.a12 = a1 + a2;
    const double emax = std::sqrt(std::max(l1, l2));
    const double n1[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2],
                          e2[0] * e1[1] - e2[1] * e1[0]};
    const double nn = std::sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
    m.bad = !(m.rho > 0x1.2p-10 * emax) || !(nn > 0.0) || !std::isfinite(nn) || !std::isfinite(m.rho);
    if (m.bad) {
      always = true;
      continue;
    }
    for (int a = 0; a < 3; a++) m.nh[a] = n1[a] / nn;
    // tau = 3.2u (10.04 |tv||e2| + 5.04 |tv||e1| + 20.1 |e1||e2|) emax / rho (1-norms), rt_brute.h
    // (divided by this node's slack factor below)
    const double k = 3.2 * u * emax / (m.rho / 1.00002) / nn * 1.0001;
    m.b1 = k * (10.04 * a2 + 5.04 * a1);
    m.b0 = k * 20.1 * a1 * a2;
    for (int a = 0; a < 3; a++) { // box of the vertices
      const double x0 = t.v0[a], x1 = x0 + e1[a], x2 = x0 + e2[a];
      lo[a] = std::min(lo[a], std::min(x0, std::min(x1, x2)));
      hi[a] = std::max(hi[a], std::max(x0, std::max(x1, x2)));
      m.vtx[0][a] = x0;
      m.vtx[1][a] = x1;
      m.vtx[2][a] = x2;
    }
  }
  DevTriGroup g;
  std::memset(&g, 0, sizeof(g));
  g.always = 1.f;
  g.rgeo = 0.f;
  g.slack = 1.f;
  if (always || count == 0) { // the sweeps open it whatever the ray: the other fields are unused
    if (count > 0) {
      g.cx = tri[(size_t)order[0]].v0[0];
      g.cy = tri[(size_t)order[0]].v0[1];
      g.cz = tri[(size_t)order[0]].v0[2];
    }
    return g;
  }
  auto up = [](double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, HUGE_VALF);
    return f;
  };
  g.cx = (float)(0.5 * (lo[0] + hi[0]));
  g.cy = (float)(0.5 * (lo[1] + hi[1]));
  g.cz = (float)(0.5 * (lo[2] + hi[2]));
  const double C[3] = {g.cx, g.cy, g.cz};
  double rg = 0, rext = 0, b0 = 0, b1 = 0, ax[3] = {0, 0, 0};
  // this node's slack factor k (statement (K)): up to slack_cap, and no larger than what doubles
  // the node's tight radius -- k rho_max <= max_v |v - C|
  double slack_k = 1.0;
  if (slack_cap > 1.0) {
    double tight = 0, rho_max = 0;
    for (const M &m : ms) {
      rho_max = std::max(rho_max, m.rho);
      for (int v = 0; v < 3; v++) {
        const double d[3] = {m.vtx[v][0] - C[0], m.vtx[v][1] - C[1], m.vtx[v][2] - C[2]};
        tight = std::max(tight, std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
      }
    }
    slack_k = std::min(slack_cap, std::max(1.0, tight / rho_max));
    slack_k = (double)(float)slack_k; // as stored
  }
  for (const M &m : ms) {
    // (S_t): the accepted hit point lies within rho_t of the triangle, hence -- the triangle being
    // the convex hull of its vertices -- within rho_t + max_v |v - C| of C
    double far_v = 0;
    for (int v = 0; v < 3; v++) {
      const double d[3] = {m.vtx[v][0] - C[0], m.vtx[v][1] - C[1], m.vtx[v][2] - C[2]};
      far_v = std::max(far_v, std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
    }
    rg = std::max(rg, far_v + slack_k * m.rho);
    rext = std::max(rext, std::fabs(m.ext[0] - C[0]) + std::fabs(m.ext[1] - C[1]) +
                              std::fabs(m.ext[2] - C[2]) + m.a12);
    b0 = std::max(b0, m.b0 / slack_k);
    b1 = std::max(b1, m.b1 / slack_k);
    const double sgn = (m.nh[0] * ms[0].nh[0] + m.nh[1] * ms[0].nh[1] + m.nh[2] * ms[0].nh[2]) < 0 ? -1.0 : 1.0;
    for (int a = 0; a < 3; a++) ax[a] += sgn * m.nh[a];
  }
  const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (!(an > 1e-3) || !std::isfinite(rg) || !std::isfinite(rext)) return g; // no cone: always open
  for (int a = 0; a < 3; a++) ax[a] /= an;
  g.ax = (float)ax[0];
  g.ay = (float)ax[1];
  g.az = (float)ax[2];
  // sines against the axis AS STORED (fp32, not exactly unit: normalise in double)
  const double fa[3] = {g.ax, g.ay, g.az};
  const double fn = std::sqrt(fa[0] * fa[0] + fa[1] * fa[1] + fa[2] * fa[2]);
  double smax = 0;
  for (const M &m : ms) {
    const double c[3] = {fa[1] * m.nh[2] - fa[2] * m.nh[1], fa[2] * m.nh[0] - fa[0] * m.nh[2],
                         fa[0] * m.nh[1] - fa[1] * m.nh[0]};
    smax = std::max(smax, std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) / fn);
  }
  g.smax = up(smax * (1.0 + 1e-6) + 1e-7);
  g.rgeo = up(rg * (1.0 + 0x1p-40));
  g.rext = up(rext * (1.0 + 1e-6));
  g.b0 = up(b0);
  g.b1 = up(b1);
  g.slack = (float)slack_k;
  g.always = 0.f;
  return g;
}

} // namespace esc
