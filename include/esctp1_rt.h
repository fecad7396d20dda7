/*
 * esctp1_rt.h -- C ABI of the MI355X-native renderer that replaces the per-pixel render
 * loop of pg42819/EscTp1RayTracer.
 *
 * Plain C: pointers, sizes and PODs only (no C++ / torch types), so the reference's
 * C++ host (or any FFI: ctypes, cgo, JNI) can bind it.  Each entry point cites the
 * reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - every function returning int returns ESC_OK (0) or a negative ESC_ERR_* code;
 *     esc_last_error() gives the thread-local message.  No exception crosses this ABI
 *     (the reference throws std::runtime_error, main.cpp:490-534, sceneloader.cpp:27-30).
 *   - images are interleaved RGB, pixel (w,h) at index (h*W + w)*3, h = 0 is the BOTTOM
 *     row (main.cpp:784-788 `image[h*W+w]`, flat form main.cpp:667-673).
 *   - all host buffers are caller-owned; the library keeps no pointer after a call returns
 *     unless stated.
 *   - one esc_context per host thread / per GPU; calls on one context are not re-entrant.
 */
#ifndef ESCTP1_RT_H
#define ESCTP1_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  ESC_OK = 0,
  ESC_ERR_INVALID = -1, /* bad argument / shape mismatch */
  ESC_ERR_HIP = -2,     /* HIP runtime error (message carries hipGetErrorString) */
  ESC_ERR_IO = -3,      /* file could not be opened / written */
  ESC_ERR_PARSE = -4,   /* OBJ/MTL rejected (mirrors sceneloader.cpp:27-30,67-70 throws) */
  ESC_ERR_NOMEM = -5,
  ESC_ERR_NO_DEVICE = -6, /* no usable gfx950 device: the product has NO CPU fallback */
  ESC_ERR_RCCL = -7       /* RCCL missing or an ncclResult_t != ncclSuccess (message carries
                             ncclGetErrorString) */
};

const char *esc_last_error(void);
/* "esctp1raytracer_amd <ver> gfx950 hip" */
const char *esc_version(void);

/* ------------------------------------------------------------------------------------
 * Host scene  == tracer::scene (src/scene/scene.h:8-44)
 * ---------------------------------------------------------------------------------- */
typedef struct esc_scene esc_scene;

/* tracer::scene::Material, scene.h:11-18, as 13 floats: ka[3] kd[3] ks[3] ke[3] Ns */
#define ESC_MATERIAL_FLOATS 13

esc_scene *esc_scene_new(void);
void esc_scene_free(esc_scene *scene);

/* Append one tracer::scene::Geometry (scene.h:20-31): de-indexed vertices, optional
 * per-vertex normals (n_normals == 0 => none), face_index triples into `vertex`.
 * The geometry is a light source iff dot(ke,ke) > 0 (sceneloader.cpp:63-64,102-104).
 * Coordinates must be finite and face indices in range (ESC_ERR_INVALID otherwise).
 * Returns the new geomID (>= 0) or a negative error. */
int esc_scene_add_geometry(esc_scene *scene, const float *vertex, int32_t n_vertices,
                           const float *normals, int32_t n_normals, const uint32_t *face_index,
                           int32_t n_faces, const float material[ESC_MATERIAL_FLOATS]);

/* EXTENSION (not in the reference): analytic spheres, cx cy cz r each,
 * one material per sphere.  Tie order: after every triangle, then by sphere index. */
int esc_scene_add_spheres(esc_scene *scene, const float *spheres_xyzr, const float *materials,
                          int32_t n_spheres);

/* model::loadobj (src/scene/sceneloader.h:10, sceneloader.cpp:14-106): OBJ + MTL ->
 * one geometry per OBJ shape, vertices de-indexed 3 per face, material = first face's
 * material, normals normalised on load.  Appends to `scene`. */
int esc_scene_load_obj(esc_scene *scene, const char *obj_path);

/* Synthetic scenes of BASELINE.json's configs (generator splitmix64):
 *   "c2" 100 spheres, "c3" 1k spheres, "c4" 10k spheres, "c5" 100,352-triangle heightfield,
 *   each with the 2-triangle floor and one single-triangle light.  n_override > 0 replaces
 *   the primitive count (spheres, or heightfield quads per side for c5). */
int esc_scene_synthetic(esc_scene *scene, const char *config, int32_t n_override);
/* eye / look-at of the synthetic configs: (0,3,6) -> (0,2,-8) */
void esc_synthetic_view(float eye[3], float look[3]);

/* introspection (tests, bindings) */
typedef struct {
  int32_t n_geometry;
  int32_t n_lights;
  int32_t n_triangles; /* sum of faces */
  int32_t n_spheres;
} esc_scene_info;
int esc_scene_get_info(const esc_scene *scene, esc_scene_info *info);
/* counts[3] = n_vertices, n_normals, n_faces */
int esc_scene_geometry_counts(const esc_scene *scene, int32_t geom, int32_t counts[3]);
int esc_scene_geometry_copy(const esc_scene *scene, int32_t geom, float *vertex, float *normals,
                            uint32_t *face_index, float material[ESC_MATERIAL_FLOATS]);
int esc_scene_light_sources(const esc_scene *scene, int32_t *geom_ids /* [n_lights] */);
int esc_scene_spheres_copy(const esc_scene *scene, float *spheres_xyzr, float *materials);

/* ------------------------------------------------------------------------------------
 * Camera == tracer::camera (src/scene/camera.h:7-41); ctor arithmetic runs on the host
 * ---------------------------------------------------------------------------------- */
typedef struct {
  float origin[3];
  float lower_left_corner[3];
  float horizontal[3];
  float vertical[3];
} esc_camera;

/* camera.h:16-29.  aspect = float(W)/H at the call site (main.cpp:548). */
void esc_camera_init(esc_camera *cam, const float lookfrom[3], const float lookat[3],
                     const float vup[3], float vfov, float aspect);

/* ------------------------------------------------------------------------------------
 * ISPC-compatible flat scene == FlatScene (src/simplify/flatten_iscp.h:9-13) with the
 * structs of src/ispc/ispc_helpers.h:16-48.  Layouts are byte-identical to what the
 * ispc compiler emits into trace_ispc.h for those declarations.
 * ---------------------------------------------------------------------------------- */
typedef struct ispc_triangle {
  float vertices[3][3];
  float normals[3][3];
  int32_t prim_id;
  int32_t geom_id;
  int32_t has_normals;
  int32_t is_light;
  float ka[3];
  float kd[3];
  float ks[3];
  float ke[3];
  float Ns;
} ispc_triangle; /* 140 bytes */

typedef struct ispc_light {
  int32_t geom_id;
  int32_t *light_faces; /* indexes into light_triangles[] */
  int32_t num_light_faces;
} ispc_light; /* 24 bytes on LP64 */

typedef struct ispc_cam {
  float lookfrom[3];
  float lookat[3];
  float vup[3];
  float vfov;
  float aspect;
} ispc_cam; /* 44 bytes */

typedef struct esc_flat_scene esc_flat_scene;
/* flatten_scene_ispc (flatten_iscp.cpp:35-111).  sort_by_centroid_x != 0 reproduces the
 * reference's std::sort at flatten_iscp.cpp:110 (it permutes primitive indices: equal-t ties and,
 * with >= 2 lights, the first occluder in index order -- quirk S3 -- can change);
 * 0 keeps (geometry, face) order == the scalar path's order and image.  Unlike the reference
 * (dangling vector, defect I4) the light_faces arrays stay valid until esc_flat_free. */
int esc_flatten_ispc(const esc_scene *scene, int32_t sort_by_centroid_x, esc_flat_scene **out);
void esc_flat_free(esc_flat_scene *flat);
ispc_triangle *esc_flat_triangles(esc_flat_scene *flat, int32_t *n);
ispc_triangle *esc_flat_light_triangles(esc_flat_scene *flat, int32_t *n);
ispc_light *esc_flat_lights(esc_flat_scene *flat, int32_t *n);
/* new_ispc_cam (flatten_iscp.cpp:117-128) */
void esc_new_ispc_cam(ispc_cam *cam, const float lookfrom[3], const float lookat[3],
                      const float vup[3], float vfov, float aspect);

/* ------------------------------------------------------------------------------------
 * THE DROP-IN: same symbol, argument list and image convention as the ISPC export
 *   src/ispc/trace.ispc:86-92, called at src/main.cpp:619-624
 * (the generated header declares it `extern "C"` inside namespace ispc with `ispc_cam &`;
 * a C++ reference is a pointer in the C ABI).  Semantics = the reference's SCALAR path
 * (main.cpp:698-791) on the flat arrays in the order given; the image is OVERWRITTEN (the
 * reference accumulates into uninitialised memory, defect I3).  Synchronous; renders on
 * device 0 (or $ESC_DEVICE); errors are reported on stderr and leave the image zeroed,
 * because the replaced function returns void.  Multi-face lights use the counter-based
 * face choice below with seed 0.  $ESC_TRACE_STAGE=bvh renders through the opt-in
 * acceleration structure (ESC_STAGE_BVH below) -- same image.
 * The reference's C++ host gets the `ispc::trace(..., ispc_cam &, ...)` form of this
 * declaration from include/trace_ispc.h (the replacement for the ISPC-generated header).
 * ---------------------------------------------------------------------------------- */
#ifndef ESC_NO_TRACE_DECL /* include/trace_ispc.h declares it with `ispc_cam &` in namespace ispc */
void trace(int32_t image_width, int32_t image_height, ispc_cam *cam, int32_t num_triangles,
           ispc_triangle triangles[], int32_t num_lights, ispc_light lights[],
           int32_t num_light_triangles, ispc_triangle light_triangles[], float *return_image,
           int32_t debug, int32_t test);
#endif

/* ------------------------------------------------------------------------------------
 * Extended entry points (persistent device state, row bands, device-resident output)
 * ---------------------------------------------------------------------------------- */
typedef struct esc_context esc_context;

enum {
  ESC_FACE_FIXED = 0, /* faceID = fixed_face: the reference's behaviour for 1-face lights */
  ESC_FACE_HASH = 1   /* faceID = splitmix64(seed,pixel,light) % n_faces; replaces the
                         std::random_device-seeded mt19937 draw of main.cpp:587-588,743-747 */
};

enum {
  ESC_STAGE_AUTO = 0, /* fastest measured variant per pass */
  ESC_STAGE_SMEM = 1, /* primitives broadcast through the scalar cache into SGPRs */
  ESC_STAGE_LDS = 2,  /* primitives staged in LDS chunks by the workgroup: north_star's sketch, kept as
                         the REFERENCE-ARITHMETIC A/B PATH -- every (ray, primitive) pair runs the
                         reference's operations, no filters, no groups, no lists.  Not a performance
                         path: 20 % behind the scalar-cache staging when both ran that arithmetic
                         (round 1), one to two orders of magnitude behind the default today
                         (profiles/r03_final/stage_lds.txt).  What it is for: an independent
                         cross-check in the tests (same image, the reference's any-hit count). */
  ESC_STAGE_BVH = 3   /* opt-in acceleration structure (what the reference's --bvh flag meant to
                         be, main.cpp:98-171,331-415): screen-space bins for primary rays,
                         light-space bins for shadow rays and a bounding-volume tree behind both
                         cull primitives before the same exact tests run, so the image is the
                         brute-force image.  One kernel per frame.  Proven bounds
                         only: triangle meshes go through the default path's lists and groups
                         unless ESC_RENDER_BVH_HEURISTIC_PADS asks for the tree.
                         Never chosen by AUTO. */
};

typedef struct {
  int32_t shadows;   /* 1 = occlusion() evaluated (main.cpp:772); 0 = "primary rays only" */
  int32_t face_mode; /* ESC_FACE_* */
  int32_t fixed_face;
  int32_t stage; /* ESC_STAGE_* */
  uint64_t seed;
  int32_t pixels_per_lane; /* 0 = auto; 1, 2 or 4 pixels carried by each work-item of the primary
                              pass (same row, 16 columns apart).  Purely a scheduling choice:
                              results are bit-identical for every value.  Honoured by ESC_STAGE_LDS;
                              ESC_STAGE_SMEM / AUTO always carry 2 (the variant the packed filter
                              bodies are written for; the others spilled and were removed). */
  int32_t flags; /* ESC_RENDER_*; 0 = defaults */
} esc_render_options;

enum {
  /* Brute force evaluates a cheap conservative FILTER per (ray, sphere) and runs the reference
   * arithmetic only where the filter cannot rule a hit out (csrc/rt_brute.h "FILTERS"); the image
   * is the same bit for bit.  This flag (or $ESC_FILTER=0) runs the reference arithmetic for
   * every pair instead -- the round-1 kernels, kept as the A/B and as a cross-check in tests. */
  ESC_RENDER_EXACT_ONLY = 1,
  /* Record HIP events on the context's stream around and between the frame's two kernels
   * (k_primary, k_shade); esc_last_kernel_ms reads them.  Brute-force stages only: under
   * ESC_STAGE_BVH the frame is one kernel and ms[0] is 0. */
  ESC_RENDER_TIME_KERNELS = 2,
  /* The reference tests primitives in index order: closest hit with a strict `t2 < *t` (the lower
   * index keeps an equal t, main.cpp:176-192), occlusion() returning at the first hit
   * (main.cpp:314-329).  By default the brute-force kernels test in ANOTHER order wherever that
   * cannot be observed: from 64 spheres / 64 triangles up they sweep bounding spheres (and normal
   * cones) of spatial groups of 8 and open only the groups a ray of the wavefront may touch
   * (csrc/rt_device.h SphGroups / TriGroups) -- closest hits with the tie rule restated on the
   * original index; shadow rays of the LAST light stop at any occluder; shadow rays of earlier
   * lights, whose FIRST occluder in index order moves the next light's ray (quirk S3), visit
   * every group and keep the accepted primitive with the lowest original index.  Shorter sphere
   * lists of the last light are swept by decreasing solid angle.  Same image bit for bit, far fewer tests;
   * esc_counters.anyhit_tests then counts what THIS order executed.  This flag switches all of
   * that off: index order everywhere, anyhit_tests == the reference's count. */
  ESC_RENDER_INDEX_ORDER = 4,
  /* The shading pass has two forms with the same arithmetic: fused (one kernel, rays re-packed
   * inside each workgroup) and queue (one launch per segment of the primitive list, rays compacted
   * across the whole band).  By default the queue form is used for long lists (>= 2,048
   * primitives) on large bands (>= 1.5 M pixels) when the lists are swept linearly -- several
   * lights, or ESC_RENDER_INDEX_ORDER; a single light's grouped lists always take the fused form,
   * the only one that sweeps groups.  These force one. */
  ESC_RENDER_SHADE_QUEUE = 8,
  ESC_RENDER_SHADE_FUSED = 16,
  /* Primary rays of grouped tables do not sweep the group levels: per camera and band the library
   * lists, for every 32 x 4 pixel tile, the primitives its rays can touch (the projection of each
   * sphere grown by what the reference's rounding can reach, of each triangle dilated in its plane
   * by its bounding radius, plus the tiles a triangle's "nearly parallel" band crosses:
   * csrc/rt_lists.h), and a wavefront tests its tile's primitives directly.  Same filters and
   * reference arithmetic behind the lists, same image.  This flag (or $ESC_LISTS=0) keeps the
   * three-level sweep of round 2 for every tile -- the A/B switch and a cross-check in tests; tiles
   * whose lists overflow (more than 512 primitives) take that sweep anyway.  The lists cost 2 KB of
   * device memory per tile and primitive kind the scene has: 133 MB for a 4K frame, 531 MB at 8K. */
  ESC_RENDER_NO_TILE_LISTS = 32,
  /* The same from the light's end: a light that offers one sample point this frame (a one-face
   * light, or ESC_FACE_FIXED; the first 4 such lights) has all its shadow rays on lines through that
   * point, so the spheres / triangles a ray can reach are listed (as pair records) per direction cell
   * of a cube map around the point, once per scene and sample point (csrc/rt_lists.h "Light lists",
   * 25 MB per light and kind); a wavefront looks its rays' cells up and tests those lists -- inside
   * the wavefront, without the workgroup-wide re-packing of the sweeps, when every primitive kind of
   * the scene is covered.  This flag (or $ESC_LLISTS=0) keeps the three-level group sweep for every
   * shadow ray; rays the lists cannot serve (an overflowing cell, an origin outside the scene box)
   * take it anyway. */
  ESC_RENDER_NO_LIGHT_LISTS = 64,
  /* The default frame of the scalar-cache staging is ONE kernel (k_frame: closest hit and shading
   * of a 64 x 8 tile; no hit planes through HBM).  This flag (or $ESC_FRAME=2) keeps the two kernels
   * of rounds 1-2, k_primary + k_shade -- the A/B switch, and what ESC_RENDER_TIME_KERNELS and the
   * queue form of the shading pass use anyway.  Same arithmetic, same image. */
  ESC_RENDER_TWO_KERNELS = 128,
  /* ESC_STAGE_BVH culls with PROVEN bounds only: spheres through its tree and bins (box pads from
   * the error bound of the discriminant), triangle meshes (more than 4 triangles) through the tile /
   * light lists and group levels of the default path, whose reach statements cover the reference's
   * rounding-noise accepts for rays that graze a triangle's plane.  This flag sends triangle meshes
   * through the tree as well: its triangle boxes carry a HEURISTIC pad (2^-12 of the scene's
   * scale) -- faster on large meshes, every test and hunt so far bit-identical, but for rays within
   * ~1e-3 rad of a triangle's plane whose rounding-noise hit lies further than the pad outside the
   * triangle it may cull a hit the reference reports. */
  ESC_RENDER_BVH_HEURISTIC_PADS = 256,
  /* The ray counters (esc_counters: instrumentation the reference does not have) are not updated by
   * this call.  Counting costs two workgroup barriers, a handful of LDS and global atomics per
   * workgroup and a wave reduction per light and pixel: 9 % of a c4 frame.  Every frame of a fixed
   * scene, camera and option set counts the same rays, so a caller that wants both (bench.py) counts
   * one frame and times the others. */
  ESC_RENDER_NO_COUNTERS = 512
};

typedef struct {
  uint64_t primary_rays; /* pixels rendered */
  uint64_t hit_pixels;   /* primary rays that hit something */
  uint64_t shadow_rays;  /* occlusion() calls (main.cpp:772): hit pixels x lights */
  uint64_t anyhit_tests; /* primitive tests those calls execute: up to and including the first
                            occluder met, else every primitive.  Equal to the reference's count
                            whenever the sweep is in index order (always with
                            ESC_RENDER_INDEX_ORDER; see there).  Group sweeps: per ray, the
                            super-group tests up to its occluder's (or all) plus 8 per group or
                            super-group opened on its behalf.  Under ESC_STAGE_BVH: the tests the
                            tree walk left for still-undecided rays */
  uint64_t anyhit_lane_tests; /* any-hit tests the GPU actually spent lanes on (64 per wave per
                                 primitive swept, decided or idle lanes included); the ratio
                                 anyhit_tests / anyhit_lane_tests is the lane efficiency of the
                                 shadow pass (SMEM stage only, 0 otherwise) */
} esc_counters;

/* Creates a context on HIP device `device` with its own stream.  Fails with
 * ESC_ERR_NO_DEVICE when there is no GPU: there is no CPU fallback. */
int esc_context_create(int32_t device, esc_context **out);
void esc_context_destroy(esc_context *ctx);
/* Launch on the caller's hipStream_t instead (e.g. a torch stream's handle). */
int esc_context_set_stream(esc_context *ctx, void *hip_stream);
void *esc_context_stream(esc_context *ctx);
int esc_context_synchronize(esc_context *ctx);

/* == flatten + hipMemcpy: stages the scene as SoA tables in HBM (replaces
 * flatten_scene_ispc's role at main.cpp:591-605).  Re-upload replaces the previous one. */
int esc_upload_scene(esc_context *ctx, const esc_scene *scene);
/* same, from ISPC flat arrays (what trace() does internally) */
int esc_upload_flat(esc_context *ctx, int32_t num_triangles, const ispc_triangle *triangles,
                    int32_t num_lights, const ispc_light *lights, int32_t num_light_triangles,
                    const ispc_triangle *light_triangles);

/* Host-only check of the arrays `trace` / esc_upload_flat would stage (no GPU needed): geom ids
 * >= 0, every light with >= 1 face and a non-null light_faces whose entries index
 * light_triangles[].  It cannot detect a DANGLING light_faces pointer -- the reference's own
 * flatten_scene_ispc leaves one (flatten_iscp.cpp:39,103); use esc_flatten_ispc instead. */
int esc_check_flat(int32_t num_triangles, const ispc_triangle *triangles, int32_t num_lights,
                   const ispc_light *lights, int32_t num_light_triangles,
                   const ispc_triangle *light_triangles);

/* == the row loop main.cpp:628-636 restricted to rows [row_begin,row_end) of a W x H frame
 * (scan_row, main.cpp:28-30, is the row-granular seam).  Asynchronous on the context's
 * stream.  Outputs are DEVICE pointers, band-local: pixel (w,h) at ((h-row_begin)*W+w)*3.
 *   d_rgb_f32  fp32 RGB accumulators (may be NULL)
 *   d_rgb_u8   8-bit RGB after the PPM clamp/truncate of main.cpp:676-682 (may be NULL)
 * Counters accumulate on the device; read them with esc_read_counters (synchronises). */
int esc_render_rows(esc_context *ctx, const esc_camera *cam, int32_t W, int32_t H,
                    int32_t row_begin, int32_t row_end, const esc_render_options *opts,
                    float *d_rgb_f32, uint8_t *d_rgb_u8);
/* Multi-GPU partition: the image is cut into strips of `strip_rows` rows counted from h = 0
 * (strip k = rows [k*strip_rows, (k+1)*strip_rows), the last one possibly short); this call
 * renders strips first_strip, first_strip + strip_stride, ... -- i.e. rank r of N calls it
 * with (first_strip = r, strip_stride = N).  Dealing strips round-robin balances sky rows
 * (primary rays only) against floor rows (primary + shadow).  strip_rows must be a multiple of
 * 8.  Output: the rendered rows packed in ascending h, esc_strip_local_rows() of them. */
int esc_render_strips(esc_context *ctx, const esc_camera *cam, int32_t W, int32_t H,
                      int32_t strip_rows, int32_t first_strip, int32_t strip_stride,
                      const esc_render_options *opts, float *d_rgb_f32, uint8_t *d_rgb_u8);
/* number of rows the call above renders (>= 0), or ESC_ERR_INVALID */
int esc_strip_local_rows(int32_t H, int32_t strip_rows, int32_t first_strip,
                         int32_t strip_stride);
/* A recorded frame: the launches of one esc_render_strips call captured into a HIP graph, replayed
 * with ONE host call per frame.  At 8 GPUs a rank's share of a 4K frame is a few tens of
 * microseconds of GPU work, the same order as the host side of a plain frame (parameter block, two
 * or more kernel launches); a recorded frame costs one hipGraphLaunch.  esc_frame_record renders one
 * plain frame first (it builds everything the kernels read), then captures the second.  A recorded
 * frame replays exactly those launches -- same camera, band, options and output buffers -- and is
 * only valid while the context's per-camera / per-scene device state stands: esc_frame_launch
 * returns ESC_ERR_INVALID once the context has rendered another camera, size or scene, or had a
 * scene uploaded (record again).  Counters accumulate as for plain frames.
 * ESC_RENDER_TIME_KERNELS cannot be recorded. */
typedef struct esc_frame esc_frame;
int esc_frame_record(esc_context *ctx, const esc_camera *cam, int32_t W, int32_t H, int32_t strip_rows,
                     int32_t first_strip, int32_t strip_stride, const esc_render_options *opts,
                     float *d_rgb_f32, uint8_t *d_rgb_u8, esc_frame **out);
int esc_frame_launch(esc_frame *frame); /* asynchronous, on the context's stream */
void esc_frame_destroy(esc_frame *frame);

/* After a gather of N such buffers to one device (block r at d_gathered + r*rank_pitch_bytes):
 * writes the H x W frame in (h*W+w) order.  bytes_per_pixel = 12 (fp32 RGB) or 3 (u8 RGB).
 * Asynchronous on the context's stream. */
int esc_assemble_strips(esc_context *ctx, const void *d_gathered, int32_t n_ranks,
                        size_t rank_pitch_bytes, int32_t W, int32_t H, int32_t strip_rows,
                        int32_t bytes_per_pixel, void *d_frame);
/* ---- acceleration structure (ESC_STAGE_BVH) -------------------------------------------
 * Built on the host from the uploaded scene the first time a frame asks for ESC_STAGE_BVH (or by
 * esc_build_accel), rebuilt when the camera leaves the region its conservative box pads were
 * computed for.  The reference times its tree build apart from the render as well
 * (main.cpp:569-579). */
typedef struct { /* 64 bytes; boxes of both children live in the parent */
  float lo0[3], hi0[3];
  float lo1[3], hi1[3];
  int32_t child[2];   /* >= 0 node index; < 0 leaf: ~block */
  uint32_t minkey[2]; /* smallest primitive key (triangles, then spheres) below each child */
} esc_bvh_node;
typedef struct {
  int32_t tri_nodes, tri_blocks, tri_depth, tri_root; /* blocks of 2 triangles */
  int32_t sph_nodes, sph_blocks, sph_depth, sph_root; /* blocks of 4 spheres */
  float build_ms; /* host build + upload, last build */
  int32_t builds; /* how many times this context has built */
  int32_t reserved[2];
} esc_accel_info;
int esc_build_accel(esc_context *ctx, const float origin[3]);
int esc_get_accel_info(esc_context *ctx, esc_accel_info *out);
/* Host only (no GPU): the same builder over a scene, for inspection and tests.  which = 0
 * triangles, 1 spheres.  Any output pointer may be NULL; counts come back in `info`.
 * prim_boxes receives the padded box of every primitive (lo xyz, hi xyz). */
int esc_scene_build_accel(const esc_scene *scene, const float origin[3], int32_t which,
                          esc_accel_info *info, esc_bvh_node *nodes, int64_t nodes_cap,
                          int32_t *order, int64_t order_cap, float *prim_boxes,
                          int64_t prim_boxes_cap);

/* Host only: the segments the queue form of the shading pass cuts occlusion()'s primitive list
 * into (main.cpp:314-329 order: triangles, then spheres).  segments receives 4 ints per segment:
 * first triangle, triangle count, first sphere PAIR record, pair-record count (either count may
 * be 0).  Returns the number of segments (<= capacity) or a negative error.  For inspection and
 * tests: every primitive must be covered exactly once, in order. */
int esc_queue_schedule(int32_t n_triangles, int32_t n_spheres, int32_t *segments,
                       int32_t capacity);

/* Host only, for inspection and tests: the static record of `count` triangles taken as ONE group
 * of the brute-force kernels' triangle groups (csrc/rt_device.h DevTriGroup): v0e1e2 holds 9
 * floats per triangle (vert0, vert1 - vert0, vert2 - vert0); record receives 12 floats:
 * centre xyz, rgeo, cone axis xyz, smax, rext, b0, b1, always.  Returns 0 or a negative error. */
int esc_tri_group_record(const float *v0e1e2, int32_t count, float record[12]);

/* The same for spheres (csrc/rt_device.h DevSphGroup): cxyzr2 holds 4 floats per sphere (centre,
 * r^2); record receives centre xyz and rgeo >= r_i + |c_i - centre| for every sphere. */
int esc_sphere_group_record(const float *cxyzr2, int32_t count, float record[4]);

/* Host only, for inspection and tests: the geometry behind the tile lists of the primary pass
 * (csrc/rt_tile_math.h, the code the binning kernels run).  esc_tile_rect: the pixel rectangle
 * rect = {w0, w1, h0, h1} (inclusive, clipped to the image) outside which no primary ray's line
 * passes within `radius` of `centre`; returns 1 (rectangle), 2 (wholly off screen), 0 (unbounded:
 * the camera plane cuts the sphere -- such a group is tested by every tile) or a negative error.
 * esc_tile_band: 1 when some ray of the pixels [32 tile_x, 32 tile_x + 32) x [row, row + 4) may
 * have |d . normal| <= kp (`normal` a unit vector, d the reference's unit direction), else 0. */
int esc_tile_rect(const esc_camera *cam, int32_t W, int32_t H, const float centre[3], double radius,
                  int32_t rect[4]);
int esc_tile_band(const esc_camera *cam, int32_t W, int32_t H, int32_t tile_x, int32_t row,
                  const float normal[3], double kp);

/* For inspection and tests: the tile lists the last frame of this context was rendered with
 * (which = 0 spheres, 1 triangles; 2 = the light lists of the shadow pass: one "tile" per direction
 * cell, tiles_x = cells per face side, tile_rows = faces x cells per side, hdr[0] = the longest
 * face-global list).  hdr receives {global primitives, cone entries, lists-off
 * flag, tiles_x, tile_rows, list capacity, global capacity, 0}; counts (may be NULL) the appended
 * primitive count of up to `capacity` tiles -- a count above the list capacity means that tile took
 * the three-level sweep.  Returns the number of tiles, 0 when the context holds no lists, or a
 * negative error.  Synchronises the context's stream. */
int esc_tile_list_counts(esc_context *ctx, int32_t which, int32_t hdr[8], int32_t *counts,
                         size_t capacity);

/* ... and the entries of ONE tile / cell (`index` in the order of esc_tile_list_counts): slots of
 * the group-sorted tables (which = 0, 1) or pair records (2, 3).  Returns the appended count (the
 * list holds min(count, capacity of the list) entries; at most `capacity` are copied). */
int esc_tile_list_ids(esc_context *ctx, int32_t which, int64_t index, int32_t *ids, int32_t capacity);

/* Host only, for inspection and tests: the spatial order the groups are cut from (k-d median
 * splits over the points; csrc/rt_device.h SphGroups / TriGroups).  xyz holds 3 floats per point;
 * order receives a permutation of 0 .. count-1 whose consecutive runs of `run`, `big` and `huge`
 * points (each a multiple of the one before) are subtrees of the splits. */
int esc_group_order(const float *xyz, int32_t count, int32_t run, int32_t big, int32_t huge,
                    int32_t *order);
/* ms[0] = k_primary, ms[1] = k_shade of the last frame rendered with ESC_RENDER_TIME_KERNELS
 * (waits for that frame).  This is how bench.py prices each kernel against its own roof. */
int esc_last_kernel_ms(esc_context *ctx, float ms[2]);
int esc_reset_counters(esc_context *ctx);
int esc_read_counters(esc_context *ctx, esc_counters *out);

/* Whole frame into HOST memory, synchronous: render + D2H.  `image` = W*H*3 floats. */
int esc_render_frame_host(esc_context *ctx, const esc_camera *cam, int32_t W, int32_t H,
                          const esc_render_options *opts, float *image, uint8_t *rgb8);

/* Single-process multi-GPU: 8-row strips dealt round-robin over n_devices (band i renders
 * strips i, i+n, ...), every band launched before any is waited on, strips copied straight
 * into the caller's host frame.  Bands share devices when there are fewer GPUs than bands.
 * (bench.py uses one process per GPU + an RCCL gather instead.) */
int esc_render_frame_multi(const esc_scene *scene, const esc_camera *cam, int32_t W, int32_t H,
                           const esc_render_options *opts, int32_t n_devices, float *image,
                           uint8_t *rgb8, float *ms_per_device /* [n_devices] or NULL */);

/* ---- native multi-GPU with an RCCL gather ------------------------
 * What a C++ host at the reference's call site (main.cpp:619-624, rows are independent:
 * main.cpp:628-636) uses to reach all GPUs of a node from ONE process: one context per device,
 * 8-row strips dealt round-robin (the esc_render_strips partition), every device renders its
 * strips, and the ONE exchange step -- the framebuffer gather to the first device -- runs over
 * RCCL: grouped ncclSend / ncclRecv, direct peer -> root over xGMI, fp32 RGB (12 B/pixel, the
 * `trace` seam's return_image) or the PPM-quantised bytes (3 B/pixel, main.cpp:676-682).  Then
 * esc_assemble_strips lays the frame out on the first device.  RCCL is bound at run time
 * (dlopen of librccl.so; $ESC_RCCL_LIB overrides), so the library itself does not link it.
 *   use_rccl = 0 replaces the exchange by hipMemcpyPeerAsync (same layout; for hosts without RCCL).
 *   With RCCL n_devices must not exceed the device count and the ids must be distinct (one
 *   communicator rank per device).  Without it ranks may SHARE devices: device_ids may repeat, and
 *   with device_ids == NULL rank i takes device i % count -- the n-rank partition, offsets, copies
 *   and assembly then run on however many GPUs there are (how a 1-GPU box tests n = 2, 3, 8).
 *   STATE OF VERIFICATION: the n > 1 path without RCCL runs in the GPU tests (ranks sharing one
 *   device); the grouped ncclSend / ncclRecv exchange has so far only run with n = 1, where it
 *   moves nothing. */
typedef struct esc_multi esc_multi;
int esc_rccl_available(void); /* 1 / 0 (esc_last_error says why not) */
int esc_multi_create(int32_t n_devices, const int32_t *device_ids /* NULL = 0..n-1 (RCCL) / i % count */,
                     int32_t use_rccl, esc_multi **out);
void esc_multi_destroy(esc_multi *m);
int esc_multi_upload_scene(esc_multi *m, const esc_scene *scene); /* replicated on every device */
/* One frame, synchronous.  gather_u8 = 0: fp32 RGB is gathered (`image` may be set, rgb8 must be
 * NULL); 1: the quantised bytes (`rgb8` may be set, image must be NULL).  Host pointers may be
 * NULL when only the device-resident frame is wanted: *d_frame (if non-NULL) receives the
 * assembled frame's address on the first device, valid until the next call on `m`. */
int esc_multi_render(esc_multi *m, const esc_camera *cam, int32_t W, int32_t H,
                     const esc_render_options *opts, int32_t gather_u8, float *image,
                     uint8_t *rgb8, void **d_frame, float *ms_per_device /* [n] or NULL */);
/* create + upload + render + destroy in one call (the shape of esc_render_frame_multi).
 * NOTE: every call creates the contexts, uploads the scene to every device AND initialises an RCCL
 * communicator (ncclCommInitAll: tens to hundreds of milliseconds), then tears all of it down.
 * Meant for a viewer's single frame; a caller that renders in a loop keeps an esc_multi. */
int esc_render_frame_multi_rccl(const esc_scene *scene, const esc_camera *cam, int32_t W, int32_t H,
                                const esc_render_options *opts, int32_t n_devices, float *image,
                                uint8_t *rgb8, float *ms_per_device /* [n_devices] or NULL */);

/* ------------------------------------------------------------------------------------
 * PPM writer == main.cpp:658-689: "P3\nW H\n255\n", rows top-down, clamp >1, int(c*255)
 * ---------------------------------------------------------------------------------- */
int esc_write_ppm(const char *path, const float *image, int32_t W, int32_t H);
/* same text from already-quantised bytes (same (h*W+w)*3 order) */
int esc_write_ppm_u8(const char *path, const uint8_t *rgb8, int32_t W, int32_t H);
void esc_quantise(const float *image, int64_t n_values, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* ESCTP1_RT_H */
