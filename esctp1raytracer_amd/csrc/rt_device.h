// rt_device.h -- HBM layout of the staged scene and the kernel parameter block.
// Shared by rt_kernels.hip (device) and rt_capi.cpp (host staging).
//
// The reference walks AoS geometry -> face_index -> vertex for every ray
// (main.cpp:179-186).  Here every primitive is one fixed-size record in a flat table that
// a whole wavefront reads at the same address at the same time (wave-uniform), so it can
// come from the scalar cache into SGPRs or from an LDS broadcast.  Everything stored is
// either an input value or the result of the SAME fp32 operation the reference performs
// per ray (edge1 = v1 - v0 etc., ray_triangle.h:14-15), so precomputing it cannot change
// a bit of the result.
#pragma once
#include <stdint.h>

namespace esc {

// ray_triangle.h:14-15 operands, 48 B.  Used by shadow rays (origin differs per pixel)
// and by shading (main.cpp:728-731 uses the same two edge vectors).
struct alignas(16) DevTri {
  float v0[3];
  float e1[3]; // vert1 - vert0
  float e2[3]; // vert2 - vert0
  int32_t geom; // material / geometry id
  int32_t pad[2];
};

// Primary rays share one origin (camera.h:32), so tvec, qvec and dot(edge2,qvec)
// (ray_triangle.h:29,37,45) are per-triangle constants for the frame.  64 B.
struct alignas(16) DevTriP {
  float e2[3];
  float e1[3];
  float tv[3]; // orig - vert0
  float qv[3]; // cross(tvec, edge1)
  float tnum;  // dot(edge2, qvec)
  float pad[3];
};

// sphere extension: general form, r2 = r*r
struct alignas(16) DevSph {
  float cx, cy, cz, r2;
};
// primary form: oc = orig - c, cc = dot(oc,oc) - r2
struct alignas(16) DevSphP {
  float ocx, ocy, ocz, cc;
};

// Two spheres per record, component-interleaved, so each field is one aligned SGPR pair that
// a v_pk_mul_f32 / v_pk_add_f32 consumes directly: sphere 2j in the low halves, 2j+1 in the high
// halves.  Odd counts are padded with a sphere that can never be hit (cc = +inf).
struct alignas(16) DevSphPair { // general form; pad half has r2 = -inf
  float cx[2], cy[2], cz[2], r2[2];
};

// ---------------------------------------------------------------------------------------
// Conservative FILTER forms of the sphere tables (rt_brute.h "filters").  The hot loops no longer
// evaluate the reference's discriminant for every (ray, sphere): they evaluate a cheaper FMA
// expression q' that is >= 0 whenever the reference's own fp32 arithmetic could accept the pair
// (margins below; proof at the top of rt_brute.h), and only the rare batches with a candidate run
// the reference arithmetic, on the exact records above.  A filter record is NOT reference data:
// nothing computed from it reaches the image.
// ---------------------------------------------------------------------------------------
// primary rays (per frame): the hoisted oc scaled so that the test "b'^2 >= ccm" reads "|b''| >= 1":
//   ccm = cc - 2^-19 (A^2 + r2) - 2^-120,  A = |ocx| + |ocy| + |ocz|          (the margin, rt_brute.h)
//   s   = (fl(sqrt(ccm)) (1 - 2^-22) - 9u A) (1 - 2^-22)                       (rounded-down sqrt, less
//                                                                               what scaling can lose)
//   (sx, sy, sz) = oc / s,  w = 0            b'' = fma(sz, dz, fma(sy, dy, fma(sx, dx, w)))
// A sphere with ccm <= 0 or s <= 0 (the camera inside its margin) is always a candidate:
//   (sx, sy, sz) = 0,  w = 2.
// Laid out (sx, w) (sy, sz): a v_pk op may read ONE scalar pair, and the first FMA needs sx and w.
struct alignas(16) DevSphF {
  float sx, w, sy, sz;
};
// shadow rays (per scene): two spheres per record like DevSphPair, centres relative to the scene
// point `shadow_center` g: c' = fl(c - g), km = r2 - |c'|^2 + 2^-16 (|c'|^2 + r2) + tiny, rounded up.
// Pad half: c' = 0, km = -inf (q' = -inf: never a candidate).
struct alignas(16) DevSphPairF {
  float cx[2], cy[2], cz[2], km[2];
};

// ---------------------------------------------------------------------------------------
// Sphere GROUPS (rt_brute.h "Sphere groups"): one more filter level in front of the sphere filter.
// At commit the spheres are put in a spatial order (k-d median splits, host/accel_build.cpp
// group_order) and cut into runs of kSphGroup; a run's bounding sphere (C, rgeo), enlarged per
// frame by what the reference's own rounding can make a member reach, is tested in the same scaled
// form as a single sphere (DevSphF), 8 groups per step, and only the members of a group some ray
// of the wave may touch go through the per-sphere filter (and, behind it, the reference
// arithmetic on the exact records).  The closest hit does not depend on the order of the tests
// except between equal t, where the reference keeps the lower index: the sorted exact test
// restores that with the original index (`orig`).  Nothing computed from a group record reaches
// the image.
// ---------------------------------------------------------------------------------------
constexpr int kSphGroup = 8;          // spheres per group
constexpr int kSphSuper = 8;          // groups per super-group
constexpr int kSphHyper = 8;          // super-groups per hyper-group (the level the sweeps start at)
constexpr int kSphGroupStep = 8;      // hyper-groups per sweep step: n_hyp is a multiple of this
constexpr int kSphGroupMinSpheres = 64;
struct alignas(16) DevSphGroup { // static: centre and the radius that holds every member sphere
  float cx, cy, cz, rgeo;        // pad group: rgeo < 0
};
struct alignas(16) DevIdx4 {
  int32_t v[4];
};
// Three levels: kSphSuper consecutive groups (64 spheres) form a super-group, kSphHyper
// consecutive super-groups (512 spheres) a hyper-group, each a subtree of the k-d order with its
// own bounding sphere in the same record forms.  A sweep tests 8 hyper-groups per step, opens the
// ones some ray of the wave may touch (one step over their 8 super-groups), and so on down.
struct SphGroups {
  int32_t n_grp, n_sup;      // n_grp = 8 n_sup, n_sup = 8 n_hyp; 0: no groups (small scenes,
  int32_t n_hyp, pad;        // ESC_RENDER_INDEX_ORDER, filters off)
  const DevSph *sorted;      // n_grp * kSphGroup spheres in group order; pad slots have r2 = -inf
  const DevSphGroup *grp;    // n_grp groups, then the n_sup super-groups, then the n_hyp hyper-groups
  const DevIdx4 *orig;       // original index of each sorted slot, 4 per record; pads INT32_MAX / 2
  DevSphP *sorted_p;         // per frame: DevSphP / DevSphF of `sorted`, DevSphF of `grp`
  DevSphF *sorted_f;
  DevSphF *grp_f;            // n_grp + n_sup + n_hyp
  // shadow rays (every light; rt_brute.h Any for the lights before the last): the same
  // sorted spheres and groups as pair tables, static per scene.  grp2_f holds the bounding spheres
  // in DevSphPairF form with R = rgeo + 0x1.6p-10 (rho_max + |C - g| + rgeo) (rt_brute.h).
  const DevSphPair *sorted2;    // n_grp * kSphGroup / 2 records
  const DevSphPairF *sorted2_f; // same
  const DevSphPairF *grp2_f;    // n_grp / 2 records (pad groups: km = -inf), then n_sup / 2, n_hyp / 2
};

// Triangles (rt_brute.h "FILTERS", triangle part).  The reference's numerators are scalar triple
// products: det = e1.(d x e2) = d.(e2 x e1), u-numerator = tv.(d x e2) = d.(e2 x tv), v-numerator =
// d.qv, so with the three vectors hoisted a (ray, triangle) pair costs three FMA dot products.
// primary rays (per frame; tv = camera origin - v0):
//   n1 = e2 x e1, n2 = e2 x tv, n3 = qv (the exact table's), M = 2^-17 |e1||e2| (|e1||e2| + |tv||e2| + |qv|)
//   (1-norms), laid out as SGPR pairs: (n1x,n1y) (n1z,n2x) (n2y,n2z) (n3x,n3y) (n3z,M) (pad,pad)
struct alignas(16) DevTriF {
  float n1[3], n2[3], n3[3], M, pad[2];
};
// shadow rays (per scene): TWO triangles per record, field-interleaved (triangle 2j in the low
// halves), relative to the scene point g: k1 = e1 x (v0 - g), k2 = e2 x (v0 - g).  With
// m = (o - g) x L per ray:  u-numerator = e2.m - L.k2,  v-numerator = L.k1 - e1.m,  det = L.n1.
// M = 2^-17 |e1||e2| (|e1||e2| + (|e1|+|e2|)(|v0-g| + rho_max)), rho_max = RenderParams::shadow_rho_max
// (rays that start further than rho_max from g take the exact path).  Pad half: all zeros, M = -1
// (A = -1 < 0: never a candidate).
struct alignas(16) DevTriPairF {
  float n1x[2], n1y[2], n1z[2], e1x[2], e1y[2], e1z[2], e2x[2], e2y[2], e2z[2];
  float k1x[2], k1y[2], k1z[2], k2x[2], k2y[2], k2z[2], M[2];
};

// Triangle PRE-filters (rt_brute.h "Triangle pre-filter"): one level above the filters.  A ray the
// reference accepts either passes within R of the triangle's centroid G or is nearly parallel to
// its plane (|det| < tau); R and tau are chosen together so that the rounding noise of the
// accepted (u, v) stays within one bounding radius of the triangle whenever |det| >= tau.  Both
// tests are 3-FMA dot products against hoisted, pre-scaled vectors:
//   |b''| >= 1   bounding sphere (G, R), scaled like DevSphF            (sx, w) (sy, sz)
//   |g''| <= 1   g'' = d . (e2 x e1) / tau'                             (gx, gy) (gz, -)
// primary rays, per frame:
struct alignas(16) DevTriPF {
  float sx, w, sy, sz;
  float gx, gy, gz, pad;
};
// shadow rays, per scene, two triangles per record: the bounding spheres in DevSphPairF form
// (centre G - g, km) followed by the scaled normals
struct alignas(16) DevTriPairPF {
  float cx[2], cy[2], cz[2], km[2];
  float gx[2], gy[2], gz[2], pad[2];
};

// scene.h:11-18 Material + whether the owning geometry has normals (main.cpp:733)
struct alignas(16) DevMat {
  float ka[3];
  float kd[3];
  float ks[3];
  float ke[3];
  float Ns;
  int32_t has_normals;
  int32_t spec_free; // material_spec_free(ks, Ns): the specular term is +-0 whatever powf returns
  int32_t pad;
};
// main.cpp:782-783 adds ks * pow(dot(N, H), Ns) to the diffuse term.  With ks == (+-0, +-0, +-0) the
// product is +-0 with the sign of ks -- i.e. ks itself, bit for bit -- as soon as the power is a
// finite value >= +0 (not NaN, not -0, not inf).  rt_kernels.hip phong() establishes at run time
// that its base lies in [0.49, 1.001] (unit N and L, dot(N, L) > 0); for such a base and an exponent in
// [0, 1024] the power lies in [+0, 2.8] (it may underflow to +0, never to -0): the skip is value-
// preserving exactly for the materials flagged here.  NaN / negative / huge Ns are not flagged.
inline int32_t material_spec_free(const float ks[3], float Ns) {
  return (ks[0] == 0.f && ks[1] == 0.f && ks[2] == 0.f && Ns >= 0.f && Ns <= 1024.f) ? 1 : 0;
}

// per-triangle vertex normals, only read at shading (main.cpp:734-737)
struct DevTriN {
  float n0[3];
  float n1[3];
  float n2[3];
};
// main.cpp:728-731 hoisted (liberty 1): normalize(cross(e1, e2)) is a constant of the triangle -- the same
// fp32 operations, once per scene (k_prepare_face_normals) instead of once per hit pixel -- with the
// geometry id beside it, so the shading half reads 16 bytes per hit instead of the 48-byte record
struct alignas(16) DevTriFace {
  float n[3];
  int32_t geom;
};

// quirk S2 (main.cpp:748-754): the light sample is light.vertex[faceID] itself; the
// candidate points (first n_faces vertices of the light geometry, each "+ 0.0f") are
// stored back to back in light_points.
struct DevLight {
  int32_t first_point;
  int32_t n_faces;
};

// ---------------------------------------------------------------------------------------
// Triangle GROUPS (rt_brute.h "Triangle GROUPS"): the sphere groups' three levels for triangles.  The
// triangles are put in a spatial order (k-d splits over the centroids), cut into groups of 8,
// super-groups of kTriSuper groups and hyper-groups of kTriHyper super-groups.  A group record has
// the form of a triangle's pre-filter record (DevTriPF / DevTriPairPF): a bounding sphere that
// holds every accepted hit point of a member, and -- for the pre-filter's "nearly parallel"
// escape -- the axis a of a cone of member normals, scaled by 1 / kappa with kappa >= sin(cone
// half-angle) + max tau' / |n1|: a ray nearly parallel to a member of the cone has |d . a| <= kappa.
// Shadow rays: the static cone over all members below.  Primary rays: the cone of THIS frame,
// over the members the camera can take the escape with (DevTriEsc).  Static part per group:
// ---------------------------------------------------------------------------------------
constexpr int kTriGroup = 8;
constexpr int kTriSuper = 16;    // groups per super-group (8: c5 22.9 ms, 16: 19.1, 32: 20.1 at the time)
constexpr int kTriHyper = 8;     // super-groups per hyper-group (the level the sweeps start at)
constexpr int kTriGroupStep = 4; // hyper-groups per sweep step: n_hyp is a multiple of this
// The pre-filter's two halves trade against each other: with the escape threshold tau_t / k an
// accepted hit point lies within k rho_t of its triangle (rt_brute.h).  Upper levels have large
// bounding spheres anyway and take a large k: their "nearly parallel" bands become k times
// thinner, and a ray outside them never opens the chain below.  k is chosen per node (host,
// tri_group_bounds): what doubles the node's tight radius, at most the caps below.
constexpr float kTriSlackGroup = 1.f; // (per node as for the upper levels: c5 4.32 instead of 4.44 ms -- not worth a second form in the mirrors; 0.25: 5.56)
constexpr float kTriSlackSuper = 8.f, kTriSlackHyper = 32.f; // c5: (2,16) 5.5, (4,32) 4.9, (8,32) 4.7, (16,32) 5.5, (8,64) 5.3 ms
constexpr int kTriGroupMinTris = 64;
struct alignas(16) DevTriGroup {
  float cx, cy, cz, rgeo; // rgeo >= rho_t + |v - C| for every vertex v of every member (rho_t: the
                          // member's bounding radius about its centroid); rgeo < 0: pad group
  float ax, ay, az, smax; // unit axis; smax >= |a x n_t / |n_t|| for every member
  float rext;             // >= |v0_t - C|_1 + |e1_t|_1 + |e2_t|_1 for every member
  float b0, b1;           // tau_t / |n1_t| <= b0 + b1 |tvec_t|_1 for every member (rt_brute.h)
  float always;           // != 0: always open (a sliver among the members, or no useful cone)
  float slack;            // the node's k of statement (K): rgeo, b0, b1 are built with tau / k, k rho
  float pad[3];
};
// the frame's cone of one node (k_prepare_tri_groups / k_prepare_tri_merge): state 0 = no member
// below can take the escape, 1 = those that can have |d . a| <= beta + |d| s, 2 = nothing can be said
struct alignas(8) DevTriEsc {
  float ax, ay, az, s, beta;
  int32_t state;
};
struct TriGroups {
  int32_t n_grp, n_sup;          // n_grp = kTriSuper n_sup, n_sup = kTriHyper n_hyp; 0: no groups
  int32_t n_hyp, pad;
  const DevTri *sorted;          // n_grp * kTriGroup triangles in group order; pads are all zeros
  const DevIdx4 *orig;           // original index of each sorted slot, 4 per record
  const DevTriGroup *grp;        // n_grp groups, then n_sup super-groups, then n_hyp hyper-groups
  DevTriP *sorted_p;             // per frame: the forms of `sorted` ...
  DevTriF *sorted_f;
  DevTriPF *sorted_pf;
  DevTriPF *grp_pf;              // ... and of the groups / super-groups / hyper-groups
  DevTriEsc *esc;                // per frame: the cones, three chains of n_grp + n_sup + n_hyp nodes
                                 // (built with k = 1, kTriSlackSuper, kTriSlackHyper; a level's
                                 // record takes its own chain, whose lower levels feed it)
  // shadow rays (every light): static, two per record
  const DevTriPairPF *sorted2_pf;
  const DevTriPairF *sorted2_f;
  const DevTriPairPF *grp2_pf;   // n_grp / 2 records, then n_sup / 2, then n_hyp / 2
};

// ---------------------------------------------------------------------------------------
// Tile lists for PRIMARY rays (rt_lists.h): which primitives can the rays of one wave's pixel tile
// (32 columns x 4 rows of the band) touch at all?  All primary rays leave one point, so the
// statements the filters rest on -- "the ray's line passes within D_i of the sphere's centre", "the
// line meets the triangle's plane within rho_t of the triangle, or the ray is nearly parallel to
// that plane" -- are regions of the image: k_bin_spheres / k_bin_triangles project every primitive
// once per camera and band (in double) and append its slot in the group-sorted table to the tiles
// of that rectangle; k_bin_tri_escape adds the triangles whose "nearly parallel" band a tile's rays
// can fall into.  The primary pass then tests its tile's primitives directly -- no sweep over the
// group levels.  A list that overflows, a frame with too many "always test" primitives, a band
// that does not start on a multiple of 4 rows: the tile falls back to the three-level sweep.
// Nothing computed from a list reaches the image.
// ---------------------------------------------------------------------------------------
#ifndef ESC_TILE_CAP
#define ESC_TILE_CAP 512
#endif
constexpr int kTileListCap = ESC_TILE_CAP;   // primitives per tile (a multiple of 4)
constexpr int kTileGlobalCap = 64;  // primitives every tile tests (the camera beside them, slivers)
constexpr int kTileMaxSpan = 8192;  // tiles one primitive may be appended to before it goes global
constexpr int kTileEscCap = 4096;   // triangles the camera is nearly in the plane of (more: lists off this frame)
constexpr int kTileHdrInts = 8 + kTileGlobalCap; // [0] n global, [1] n cone entries, [2] lists off, [8..) global ids
struct alignas(16) TileEsc { // one triangle whose "nearly parallel" escape rays of this frame can take:
  double fA, fH, fV;         // rays with |p . n| <= kp |p|; p . n = fA + s fH + t fV is affine in the
  float kp;                  // image-plane coordinates (s, t) of p = A + s H + t V
  int32_t id;                // slot in the sorted table
  // ... AND |p . m_u| <= kpu |p|, |p . m_v| <= kpv |p| (rt_lists.h "escape rays pass the edges' planes
  // too"): unit normals of the planes through the ray origin and an edge line, same affine form
  double uA, uH, uV, vA, vH, vV;
  float kpu, kpv;
  int32_t pad[2];
};
struct TileLists {
  int32_t *hdr;      // kTileHdrInts
  int32_t *cnt;      // [tiles_x * tile_rows] appended ids (may exceed the cap: that tile falls back)
  int32_t *ids;      // [tiles][kTileListCap]
  TileEsc *esc;      // [kTileEscCap] (triangle groups only)
  int32_t tiles_x;   // ceil(W / 32)
  int32_t tile_rows; // ceil(n_local_rows / 4)
  int32_t enabled;   // 0: no lists (the sweep over the levels)
  int32_t pad;
};

// ---------------------------------------------------------------------------------------
// Light lists for SHADOW rays (rt_lists.h "Light lists"): the same idea from the other end.  A
// light that offers one sample point P this frame (a one-face light, or ESC_FACE_FIXED) has every
// shadow ray on a line through P, so "which spheres can this ray's line pass within reach of" is a
// region of the directions around P: a cube map of R x R cells per face, each cell holding the PAIR
// records (rt_device.h SphGroups::sorted2) with a sphere whose reach disc the cell's directions
// touch.  Built once per scene and sample point.  A wave looks its rays' cells up, tests those
// lists with the reference arithmetic and falls back to the three-level group sweep when a cell
// overflows or a ray starts outside the region the reach was computed for.
// ---------------------------------------------------------------------------------------
constexpr int kLightListMax = 4;    // lights (sample points) that get lists
#ifndef ESC_LL_RES
#define ESC_LL_RES 128
#endif
constexpr int kLightListRes = ESC_LL_RES;  // cells per cube-face side
constexpr int kLightListCap = 64;   // pair records per cell (a multiple of 4)
constexpr int kLightEscCap = 1024;  // triangles per light whose plane (nearly) holds the sample point
struct alignas(16) LightEsc { // |v . n| <= kp over a cell's directions v = +-e_m + u e_a + w e_b
  float nx, ny, nz, kp;
  int32_t pair, pad[3];
};
struct LightLists {
  int32_t *hdr;  // [n_listed * 6][kTileHdrInts]: [0] face-global pair records, [1] (face 0 of a light:)
                 // its LightEsc entries, [2] off, [8..) the face-global ids
  int32_t *cnt;  // [n_listed * 6 * R * R]
  int32_t *ids;  // [cells][kLightListCap]
  LightEsc *esc; // [n_listed][kLightEscCap] (triangle lists only)
  int32_t n_listed; // lights 0 .. n_listed - 1 have lists, built for light_points[point[li]]
  int32_t R;
  int32_t point[kLightListMax];
  int32_t enabled, pad;
};

// hand-over between k_primary and k_shade: the closest hit of every pixel of the band
// (main.cpp:715-722 state) as three planes of n_pixels dwords each, so every store / load is a
// run of consecutive dwords:
//   idx  -1 none; [0,n_tri) triangle; n_tri + k sphere k          written for every pixel
//   t    the closest t                                            written for hit pixels only
//   v    quirk S1: only v survives (main.cpp:307,310)             written for hit pixels, and only
//                                                                 when some geometry has normals
// (a 16-byte record per pixel, sky included, cost 2 x 133 MB of HBM traffic per 4K frame; the
// planes cost 2 x (33 + 0.68 x 33) MB on c4)
struct HitPlanes {
  int32_t *idx;
  float *t;
  float *v; // only touched when RenderParams::tri_n != nullptr
};

// ---------------------------------------------------------------------------------------
// Queue form of the shadow pass (rt_kernels.hip k_shadow_setup / k_anyhit_segment /
// k_shade_finish).  occlusion() (main.cpp:314-329) walks the primitive list in index order and
// stops at the first hit, so rays retire all along the list; a wave that keeps its 64 rays from
// start to end idles more and more lanes.  Here the list is cut into segments, one kernel launch
// each, and between segments the rays still looking are compacted ACROSS THE WHOLE BAND into a
// queue of pixel ids, so every wave of every segment starts full and no retired ray holds a wave
// slot.  Every ray still meets the primitives in index order: first occluder, its t2 (quirk S3)
// and the image are unchanged.
// ---------------------------------------------------------------------------------------
struct alignas(16) ShadowRay { // one per pixel of the band, indexed by band-local pixel
  float ox, oy, oz; // origin (main.cpp:757 `hit`); once decided: ox = the occluder's t2
  float tb;         // bound len - eps (main.cpp:764); 0 = no ray, or decided
  float lx, ly, lz; // unit direction (main.cpp:766)
  int32_t kocc;     // -1, or the first occluder's index in (triangles, spheres) order
};
constexpr uint32_t kQueueInvalid = 0xffffffffu; // pad entry of a partly filled 64-id chunk
struct ShadeQueue {
  ShadowRay *rays;    // [n_pixels]
  uint32_t *q[2];     // ping-pong queues of pixel ids, in chunks of 64
  uint32_t *ctl;      // per (light, segment): [0] chunks appended to the segment's OUTPUT queue,
                      // [1] work-fetch cursor; zeroed at the start of every frame
  float *state;       // n_lights > 1 only: planes t, r, g, b carried from light to light
};
struct SegArgs { // one launch of k_anyhit_segment
  const uint32_t *qin;        // nullptr: every pixel of the band in order (first segment)
  const uint32_t *in_chunks;  // chunks in qin (device); unused with qin == nullptr
  uint32_t n_identity_chunks; // chunks when qin == nullptr
  uint32_t n_pixels;
  uint32_t *qout;             // nullptr: last segment (survivors are simply not occluded)
  uint32_t *out_chunks;
  uint32_t *cursor;
  int32_t tri_first, tri_count; // triangles of this segment (tested first: index order) ...
  int32_t rec_first, rec_count; // ... then its sphere PAIR records; either count may be 0
  // the pair tables this launch sweeps: the scene's (index order) or, for the LAST light, the
  // copies sorted by how much of the light's sky each sphere covers (RenderParams::sph2_ord)
  const DevSphPair *sph2;
  const DevSphPairF *sph2_f;
};

// ---------------------------------------------------------------------------------------
// acceleration structure (ESC_STAGE_BVH; the reference's --bvh intent, main.cpp:98-171,
// aabb.cpp:67-110).  Binary BVH whose node holds the boxes of BOTH children, 64 B = one
// s_load_dwordx16: a whole wavefront walks the tree together (wave-uniform node index, node in
// SGPRs, every lane tests its own ray against the two boxes), so the walk needs no per-lane
// gathers and no per-lane stack.  Leaves are fixed-size blocks of primitives in tree order
// (2 triangles = 96 B, 4 spheres = 64 B, padded with primitives that can never be hit); `order`
// maps block slot -> original primitive index (-1 = pad).  Boxes are padded on the host
// (accel_build.cpp) so that culling can never drop a primitive the exact test would accept.
// ---------------------------------------------------------------------------------------
struct alignas(64) BvhNode {
  float lo0[3], hi0[3]; // child 0
  float lo1[3], hi1[3]; // child 1
  int32_t child[2];     // >= 0 node index; < 0 leaf: ~block
  uint32_t minkey[2];   // smallest primitive key below each child (key = index in triangles-then-
                        // spheres order, the order occlusion() meets primitives, main.cpp:314-329)
};
constexpr int kTinyTris = 4; // up to this many triangles are tested directly, no tree (and none of its pads)
constexpr int kTriBlock = 2; // triangles per leaf block
constexpr int kSphBlock = 4; // spheres per leaf block
constexpr int kBvhMaxDepth = 60; // nodes on a root-to-leaf path; the walk's stack is one VGPR (64 lanes)
struct alignas(16) TriBlock {
  DevTri t[kTriBlock];
};
struct alignas(64) SphBlock {
  DevSph s[kSphBlock];
};
struct alignas(64) TriBlockP { // per-frame primary form of a TriBlock
  DevTriP t[kTriBlock];
};
struct alignas(64) SphBlockP {
  DevSphP s[kSphBlock];
};
struct BvhRef {
  const BvhNode *nodes;
  const void *blocks;   // TriBlock[] or SphBlock[]
  const void *blocks_p; // TriBlockP[] or SphBlockP[]: the same slots, hoisted for the camera origin
  const int32_t *order; // block * block_size + slot -> original index, -1 for pads
  int32_t root;         // as BvhNode::child; meaningless when the primitive count is 0
  int32_t pad;
};

// Screen-space bins for PRIMARY rays under ESC_STAGE_BVH.  All primary rays leave one point, so
// "which primitives can the rays of this 32 x 8 pixel tile meet" is a projection: k_bin_primary
// projects every primitive's padded box (the same boxes the tree is built from) onto the image
// plane and appends the primitive to the bins its footprint overlaps.  k_primary<BVH> then tests
// just its bin with the exact tests.  A bin that overflows, or a band whose tiles do not sit on
// 8-row image boundaries, falls back to the tree walk.  Primitives that straddle the camera
// plane, or cover very many bins, go to a short "global" list every tile tests.
constexpr int kBinCap = 32;      // ids per bin and primitive kind
constexpr int kBinGlobalCap = 16; // ids of the global lists
constexpr int kBinHdrInts = 64;  // [0] n global triangles, [1] n global spheres,
                                 // [2, 2+16) global triangle ids, [18, 34) global sphere ids
struct BinGrid {
  int32_t *hdr;     // kBinHdrInts, then counts[n_bins][2] (triangles, spheres): zeroed per frame
  int32_t *tri_ids; // [n_bins][kBinCap]
  int32_t *sph_ids; // [n_bins][kBinCap]
  int32_t tiles_x, groups_y; // bins = 32-pixel columns x 8-row groups of IMAGE rows
};
struct PrimBoxDev {
  float lo[3], hi[3];
};

// The same idea for SHADOW rays: every shadow ray towards one light sample point L (quirk S2
// makes the sample a fixed point per light face) lies on a line through L, so an occluder must be
// seen from L in the direction of the shaded point.  Directions from L are binned on a cube map
// (6 faces x R x R cells); k_bin_light projects every padded primitive box onto it once per scene
// (lights do not move with the camera).  A shadow ray looks up the cell of (hit point - L) and
// tests that cell's primitives (per-lane lists) with the exact test.  Per face, primitives that straddle the
// face's plane through L go to a short face-global list; overflowing cells / lists make the rays
// that land there walk the tree instead.
constexpr int kLightGridsMax = 4; // light sample points that get a cube map (more: tree walk only)
struct LightBins {
  int32_t *face_hdr; // [n_points*6][kBinHdrInts]  (layout of BinGrid::hdr)
  int32_t *counts;   // [n_points*6*R*R][2]
  int32_t *tri_ids;  // [n_cells][kBinCap]
  int32_t *sph_ids;  // [n_cells][kBinCap]
  int32_t n_points;  // 0: no light bins
  int32_t R;
};

struct RenderParams {
  // camera.h:36-39
  float origin[3];
  float llc[3];
  float horizontal[3];
  float vertical[3];
  int32_t W, H;
  // local row lr (ascending h) -> image row h0 + (lr / strip_rows) * strip_step + lr % strip_rows
  int32_t h0, n_local_rows, strip_rows, strip_step;
  int32_t n_tri, n_sph, n_lights, n_geom;
  const DevTri *tri;
  const DevTriP *tri_p;
  const DevTriN *tri_n; // nullptr when no geometry has normals
  const DevTriFace *tri_face; // n_tri records (k_frame)
  const DevSph *sph;
  const DevSphP *sph_p;
  const DevSphPair *sph2;    // ceil(n_sph / 2) records
  const DevSphF *sph_f;      // filter form of sph_p (per frame), n_sph records
  const DevSphPairF *sph2_f; // filter form of sph2 (per scene), ceil(n_sph / 2) records
  // Last light only (its occluder's t2 and index are never read again, quirk S3 ends there): the
  // same pair tables in the order of decreasing solid angle seen from that light's first sample
  // point, so that occluded rays meet AN occluder early.  nullptr: index order everywhere.
  const DevSphPair *sph2_ord;
  const DevSphPairF *sph2_f_ord;
  const DevTriPF *tri_pf;      // pre-filter form of tri_p (per frame), n_tri records
  const DevTriPairPF *tri2_pf; // pre-filter form of tri for shadow rays (per scene), ceil(n_tri / 2)
  const DevTriF *tri_f;      // filter form of tri_p (per frame), n_tri records
  const DevTriPairF *tri2_f; // filter form of tri for shadow rays (per scene), ceil(n_tri / 2)
  float shadow_rho_max;      // 1-norm radius around g inside which DevTriPairF's margins hold
  float shadow_center[3];    // g of DevSphPairF / DevTriPairF
  float scene_lo[3], scene_hi[3]; // the scene's box, grown: shadow-ray origins the light lists serve
  int32_t use_filter;        // 0: every test runs the reference arithmetic (A/B switch, tests)
  const int32_t *sph_mat; // material index of sphere k (already offset by n_geom)
  const DevMat *mat;      // [n_geom + n_sphere_materials]
  const DevLight *lights;
  const float *light_points; // xyz0 per candidate point
  int32_t shadows, face_mode, fixed_face, pad0;
  uint64_t seed;
  float *out_f32;   // band-local, may be null
  uint8_t *out_u8;  // band-local, may be null
  // kCounterSets replicas of {primary, hit, shadow rays, any-hit tests, 4 spare}, 64 B each
  unsigned long long *counters;
  SphGroups sg;                 // sphere groups (both passes)
  TriGroups tg;                 // triangle groups (both passes)
  TileLists sl, tl;             // primary rays: tile lists of spheres / triangles
  LightLists ll;                // shadow rays: light lists of sphere pair records
  LightLists lt;                // ... and of triangle pair records
  ShadeQueue sq;                // queue form of the shadow pass (brute force, large scenes)
  HitPlanes hits;               // band-local, n_local_rows * W pixels (scratch owned by the context)
  BvhRef bvh_tri, bvh_sph;      // ESC_STAGE_BVH only
  BinGrid bins;                 // ESC_STAGE_BVH only; hdr == nullptr: no bins, walk the tree
  LightBins lbins;              // ESC_STAGE_BVH only; n_points == 0: none
};

// pixel tile of one 256-thread workgroup: 2 x 2 waves, each wave (16*PX) x 4 pixels, so the
// tile is (32*PX) x 8 with PX = pixels per lane
constexpr int kTileH = 8;
constexpr int kCounterSets = 64; // replicas of the ray counters (contention), summed on read
constexpr int kLdsChunkBytes = 32768; // LDS staging chunk (ESC_STAGE_LDS)

} // namespace esc
