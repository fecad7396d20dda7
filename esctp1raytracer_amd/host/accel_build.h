// accel_build.h -- host builder of the acceleration structure the ESC_STAGE_BVH kernels walk
// (rt_device.h BvhNode).  The reference's --bvh path (main.cpp:98-171: median split over
// triangles sorted by v0.x, aabb.cpp:67-110 slab test) is broken and is not
// followed; what is kept is its intent -- fewer primitive tests per ray, same image.
#pragma once
#include <cstdint>
#include <vector>

#include "../csrc/rt_device.h"

namespace esc {

struct PrimBox {
  float lo[3], hi[3];
};

// where ray origins can be: the camera and every surface point of the scene (shadow rays start
// on the surface that was hit, main.cpp:757-758) and, with more than one light point, a ball of
// radius `ball` around the camera (quirk S3: a later light's shadow ray starts on the primary ray
// at the previous occluder's distance, possibly outside the scene)
struct OriginBounds {
  double lo[3], hi[3];
  double ball = 0.0;
  double ball_mult = 0.0;    // ball = ball_mult x (diagonal of scene + lights + camera): light points - 1
  double cam[3] = {0, 0, 0}; // the camera the bounds were built for
  // may a frame with the camera at p use pads computed for these bounds?  (a camera that moved by
  // d sees a scene diagonal at most d longer, so its ball is at most ball + ball_mult d)
  bool contains(const float p[3]) const {
    double r = 0.0;
    if (ball > 0.0) {
      double d2 = 0.0;
      for (int a = 0; a < 3; a++) d2 += ((double)p[a] - cam[a]) * ((double)p[a] - cam[a]);
      r = ball + ball_mult * __builtin_sqrt(d2);
    }
    for (int a = 0; a < 3; a++)
      if (!(p[a] - r >= lo[a] && p[a] + r <= hi[a])) return false;
    return true;
  }
};

struct BuiltBvh {
  std::vector<BvhNode> nodes;
  std::vector<int32_t> order; // n_blocks * block entries: original primitive index, -1 = pad
  int32_t root = 0;           // >= 0 node index, < 0 ~block (whole set fits one block)
  int32_t depth = 0;          // most internal nodes on a root-to-leaf path (stack need of the walk)
  int32_t n_blocks = 0;
};

// Scene bounds united with `origin`, grown by a tenth of the diagonal.
OriginBounds origin_bounds(const std::vector<DevTri> &tri, const std::vector<DevSph> &sph,
                           const std::vector<float> &light_points_xyz0, const float origin[3]);

// Padded primitive bounds (see accel_build.cpp for the error analysis behind the pads).
void triangle_boxes(const std::vector<DevTri> &tri, const OriginBounds &ob,
                    std::vector<PrimBox> &out);
void sphere_boxes(const std::vector<DevSph> &sph, const OriginBounds &ob,
                  std::vector<PrimBox> &out);

// Binned-SAH build over the boxes; primitive i gets key key_base + i.  Leaves hold up to
// `block` primitives.  Falls back to median splits wherever SAH would let a path grow past
// max_depth, so depth <= max_depth always.
void build_bvh(const std::vector<PrimBox> &boxes, int block, uint32_t key_base, int max_depth,
               BuiltBvh &out);

// Sphere groups of the brute-force passes (rt_device.h SphGroups): a spatial order whose
// consecutive runs of `run` spheres (groups), `big` (super-groups) and `huge` (hyper-groups; each a
// multiple of the one before) are neighbours, and the bounding sphere of one run.
void group_order(const std::vector<DevSph> &sph, int run, int big, int huge,
                 std::vector<int32_t> &order);
void group_order_points(const std::vector<float> &xyz, int run, int big, int huge,
                        std::vector<int32_t> &order);
DevSphGroup group_bounds(const std::vector<DevSph> &sph, const int32_t *order, int count);
// The same for triangles (rt_device.h TriGroups): order over the centroids; a group's static record.
void group_order(const std::vector<DevTri> &tri, int run, int big, int huge,
                 std::vector<int32_t> &order);
DevTriGroup tri_group_bounds(const std::vector<DevTri> &tri, const int32_t *order, int count,
                             double slack_cap = 1.0);

} // namespace esc
