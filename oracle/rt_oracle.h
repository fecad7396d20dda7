/*
 * rt_oracle.h -- CPU restatement of the reference's scalar render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (esctp1raytracer_amd/,
 * include/, the C-ABI library, the viewer) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the reported CPU baseline.
 *
 * What it restates (reference = /root/reference, pg42819/EscTp1RayTracer):
 *   src/math/vec.h:95-139            dot / cross / +,-,*,/ / normalize / length
 *   src/scene/camera.h:16-34         pinhole camera ctor + get_ray
 *   src/scene/ray_triangle.h:7-57    Moller-Trumbore, mixed fp32/fp64
 *   src/main.cpp:176-192,302-312     closest hit over (geometry, face)
 *   src/main.cpp:314-329             any-hit occlusion (mutates t)
 *   src/main.cpp:698-791             scan_row: ray gen, normal, per-light shading
 *   src/main.cpp:658-689             clamp / int(x*255) / P3 PPM text
 *
 * Parity pin (see DESIGN.md "Oracle"): the vec / camera / intersect_triangle
 * restatements are checked bit-for-bit against the reference's own headers
 * compiled untouched into oracle/_ref/ (oracle/Makefile).  scan_row lives in
 * main.cpp, which cannot be compiled here (it includes the ISPC-generated
 * trace_ispc.h and `ispc` is not in the image; no stand-in is written), so the
 * shading restatement is pinned by the reference PPM MD5 for scene `one`
 * recorded in SURVEY.md Appendix B and by the quirk-ablation counts of
 * SURVEY.md section 8(c).  The analytic sphere primitive does not exist in the
 * reference: for spheres this oracle is the definition ("parity unpinned by the
 * reference"), semantics frozen from SURVEY.md section 8(d).
 *
 * Build strictly: gcc -O2 -ffp-contract=off -fno-fast-math (no -march=native).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scene.h:11-18 Material (lightsource flag lives in orc_scene.light_sources) */
typedef struct {
  float ka[3];
  float kd[3];
  float ks[3];
  float ke[3];
  float Ns;
} orc_material;

/* scene.h:20-31 Geometry: de-indexed vertex array + per-face index triples */
typedef struct {
  int32_t n_vertices;
  const float *vertex; /* [n_vertices][3] */
  int32_t n_normals;   /* 0 => geometry has no normals (main.cpp:733) */
  const float *normals; /* [n_normals][3] */
  int32_t n_faces;
  const uint32_t *face_index; /* [n_faces][3] */
  orc_material material;      /* object_material */
} orc_geometry;

/* scene.h:34-35 + the sphere extension */
typedef struct {
  int32_t n_geometry;
  const orc_geometry *geometry;
  int32_t n_lights;
  const int32_t *light_sources; /* geometry ids, scene.h:35 */
  int32_t n_spheres;
  const float *spheres;            /* [n_spheres][4] = cx cy cz r */
  const int32_t *sphere_material;  /* [n_spheres] index into sphere_materials */
  const orc_material *sphere_materials;
} orc_scene;

/* the four vectors camera.h:36-39 keeps for get_ray */
typedef struct {
  float origin[3];
  float lower_left_corner[3];
  float horizontal[3];
  float vertical[3];
} orc_camera;

enum {
  ORC_FACE_FIXED = 0, /* faceID = fixed_face for every pixel/light */
  ORC_FACE_HASH = 1   /* faceID = splitmix64(seed, pixel, light) % n_faces */
};

/* quirk toggles: all ON reproduces the reference.  Turning one OFF exists only
 * so tests can reproduce SURVEY.md 8(c)'s ablation counts. */
enum {
  ORC_QUIRK_S1 = 1, /* u lost: intersect() passes v for both u and v, main.cpp:307,310 */
  ORC_QUIRK_S3 = 2, /* t carried over between lights, main.cpp:757-764,772 */
  ORC_QUIRK_ALL = 3
};

typedef struct {
  int32_t shadows;   /* 1: occlusion() evaluated; 0: treated as false ("primary only") */
  int32_t face_mode; /* ORC_FACE_* */
  int32_t fixed_face;
  uint64_t seed;
  int32_t quirks; /* ORC_QUIRK_ALL for the reference's behaviour */
} orc_options;

typedef struct {
  uint64_t primary_rays; /* pixels rendered */
  uint64_t hit_pixels;   /* primary rays that hit something */
  uint64_t shadow_rays;  /* occlusion() calls (0 when shadows == 0) */
  uint64_t anyhit_tests; /* primitive tests executed inside those calls */
} orc_counters;

/* ---- pieces, exported so tests can pin them one by one ---- */
float orc_dot(const float a[3], const float b[3]);                      /* vec.h:95-101 */
void orc_cross(const float a[3], const float b[3], float out[3]);       /* vec.h:103-109 */
void orc_normalize(const float v[3], float out[3]);                     /* vec.h:135-137 */
float orc_length(const float v[3]);                                     /* vec.h:139 */

/* camera.h:16-29 */
void orc_camera_init(orc_camera *cam, const float lookfrom[3], const float lookat[3],
                     const float vup[3], float vfov, float aspect);
/* camera.h:31-34 */
void orc_camera_get_ray(const orc_camera *cam, float s, float t, float dir_out[3]);

/* ray_triangle.h:7-57; returns 1 on accept and updates *t,*u,*v */
int orc_intersect_triangle(const float orig[3], const float dir[3], const float vert0[3],
                           const float vert1[3], const float vert2[3], float *t, float *u,
                           float *v);

/* sphere extension; returns 1 on accept and updates *t */
int orc_intersect_sphere(const float orig[3], const float dir[3], const float sphere[4],
                         float *t);

/* counter-based light-face choice shared with the HIP path (ORC_FACE_HASH) */
uint32_t orc_face_hash(uint64_t seed, uint32_t pixel, uint32_t light, uint32_t n_faces);

/* ---- frame ----
 * image: W*H*3 floats, pixel (w,h) at (h*W+w)*3, h = 0 is the bottom row
 * (main.cpp:784-786, flat layout of main.cpp:667-673).  Rows [row_begin,row_end)
 * are overwritten; other rows are untouched.  n_threads >= 1 (rows are
 * independent, main.cpp:628-636). */
void orc_render(const orc_scene *scene, const orc_camera *cam, int32_t W, int32_t H,
                int32_t row_begin, int32_t row_end, const orc_options *opts, float *image,
                orc_counters *counters, int32_t n_threads);

/* Same, for an arbitrary list of image rows; row rows[i] is written to packed_rows + i*W*3.
 * Used for bounded CPU-baseline samples and for spot checks of large frames. */
void orc_render_row_list(const orc_scene *scene, const orc_camera *cam, int32_t W, int32_t H,
                         const int32_t *rows, int32_t n_rows, const orc_options *opts,
                         float *packed_rows, orc_counters *counters, int32_t n_threads);

/* main.cpp:676-682: clamp >1, int(c*255); out = W*H*3 bytes in the SAME (h*W+w) order */
void orc_quantise(const float *image, int64_t n_values, uint8_t *out);

/* main.cpp:661-685: P3 text, rows written top-down.  Returns 0 on success. */
int orc_write_ppm(const char *path, const float *image, int32_t W, int32_t H);

#ifdef __cplusplus
}
#endif
#endif
