/*
 * rt_oracle.c -- CPU restatement of the reference's scalar render path.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Plain C, strict IEEE:
 *   gcc -O2 -ffp-contract=off -fno-fast-math -std=gnu11   (never -march=native / -ffast-math:
 *   SURVEY.md Appendix A measured 1,986-3,297 flipped pixels under those flags)
 *
 * Every fp operation below is written in the order the reference evaluates it; the
 * file:line in each comment is where that order comes from (/root/reference/...).
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vec.h */

/* vec.h:95-101: sum starts at 0 and accumulates x,y,z in that order */
float orc_dot(const float a[3], const float b[3]) {
  float sum = 0;
  for (int i = 0; i < 3; i++) sum += (a[i] * b[i]);
  return sum;
}

/* vec.h:103-109 */
void orc_cross(const float a[3], const float b[3], float out[3]) {
  float d0 = a[1] * b[2] - a[2] * b[1];
  float d1 = a[2] * b[0] - a[0] * b[2];
  float d2 = a[0] * b[1] - a[1] * b[0];
  out[0] = d0;
  out[1] = d1;
  out[2] = d2;
}

static void v_add(const float a[3], const float b[3], float o[3]) { /* vec.h:111-113 */
  o[0] = a[0] + b[0];
  o[1] = a[1] + b[1];
  o[2] = a[2] + b[2];
}
static void v_sub(const float a[3], const float b[3], float o[3]) { /* vec.h:115-117 */
  o[0] = a[0] - b[0];
  o[1] = a[1] - b[1];
  o[2] = a[2] - b[2];
}
static void v_scale(const float a[3], float s, float o[3]) { /* vec.h:127-133 */
  o[0] = a[0] * s;
  o[1] = a[1] * s;
  o[2] = a[2] * s;
}
static void v_div(const float a[3], float s, float o[3]) { /* vec.h:119-125: true divides */
  o[0] = a[0] / s;
  o[1] = a[1] / s;
  o[2] = a[2] / s;
}

/* vec.h:135-137: v / sqrt(dot(v,v)); T=float selects the float sqrt overload */
void orc_normalize(const float v[3], float out[3]) { v_div(v, sqrtf(orc_dot(v, v)), out); }

/* vec.h:139 */
float orc_length(const float v[3]) { return sqrtf(orc_dot(v, v)); }

/* --------------------------------------------------------------- camera.h */

/* camera.h:16-29 */
void orc_camera_init(orc_camera *cam, const float lookfrom[3], const float lookat[3],
                     const float vup[3], float vfov, float aspect) {
  /* :19 `float theta = vfov * M_PI / 180;` -- double product and quotient, then narrowed */
  float theta = (float)((double)vfov * M_PI / 180);
  /* :20 tan(float) resolves to the float overload through <math.h> in C++ */
  float half_height = tanf(theta / 2);
  float half_width = aspect * half_height; /* :21 */
  float w[3], u[3], v[3], tmp[3], tmp2[3];
  memcpy(cam->origin, lookfrom, sizeof(float) * 3); /* :22 */
  v_sub(lookfrom, lookat, tmp);
  orc_normalize(tmp, w); /* :23 */
  orc_cross(vup, w, tmp);
  orc_normalize(tmp, u); /* :24 */
  orc_cross(w, u, v);    /* :25 */
  /* :26 ((origin - u*hw) - v*hh) - w */
  v_scale(u, half_width, tmp);
  v_sub(cam->origin, tmp, tmp2);
  v_scale(v, half_height, tmp);
  v_sub(tmp2, tmp, tmp2);
  v_sub(tmp2, w, cam->lower_left_corner);
  /* :27 (u*2.f)*hw   :28 (v*2.f)*hh */
  v_scale(u, 2.f, tmp);
  v_scale(tmp, half_width, cam->horizontal);
  v_scale(v, 2.f, tmp);
  v_scale(tmp, half_height, cam->vertical);
}

/* camera.h:31-34: normalize(((llc + horizontal*s) + vertical*t) - origin) */
void orc_camera_get_ray(const orc_camera *cam, float s, float t, float dir_out[3]) {
  float a[3], b[3];
  v_scale(cam->horizontal, s, a);
  v_add(cam->lower_left_corner, a, a);
  v_scale(cam->vertical, t, b);
  v_add(a, b, a);
  v_sub(a, cam->origin, a);
  orc_normalize(a, dir_out);
}

/* --------------------------------------------------------- ray_triangle.h */

/* ray_triangle.h:7-57 "the original jgt code", det/inv_det in double */
int orc_intersect_triangle(const float orig[3], const float dir[3], const float vert0[3],
                           const float vert1[3], const float vert2[3], float *t, float *u,
                           float *v) {
  float edge1[3], edge2[3], tvec[3], pvec[3], qvec[3];
  double det, inv_det;
  const float eps = FLT_EPSILON;

  v_sub(vert1, vert0, edge1); /* :14 */
  v_sub(vert2, vert0, edge2); /* :15 */
  orc_cross(dir, edge2, pvec); /* :18 */
  det = orc_dot(edge1, pvec);  /* :21 float dot widened */
  if (det > -eps && det < eps) return 0; /* :23-25 */
  inv_det = 1.0f / det; /* :26 double divide */
  v_sub(orig, vert0, tvec); /* :29 */
  float u2 = (float)(orc_dot(tvec, pvec) * inv_det); /* :32 double product, narrowed */
  if (u2 < eps || u2 > 1.0f) return 0; /* :33 */
  orc_cross(tvec, edge1, qvec); /* :37 */
  float v2 = (float)(orc_dot(dir, qvec) * inv_det); /* :40 */
  if (v2 < eps || u2 + v2 > 1.0f) return 0; /* :41 float add */
  float t2 = (float)(orc_dot(edge2, qvec) * inv_det); /* :45 */
  if (t2 < eps) return 0; /* :46 */
  if (t2 >= *t) return 0; /* :49 strict: first primitive wins ties */
  *t = t2; /* :52-54 */
  *u = u2;
  *v = v2;
  return 1;
}

/* ------------------------------------------------------- sphere extension */

/* SURVEY.md 8(d).  Not in the reference; this function IS the definition.
 * All fp32, reference dot order, no contraction; ray dir is unit length so a == 1. */
int orc_intersect_sphere(const float orig[3], const float dir[3], const float sphere[4],
                         float *t) {
  const float eps = FLT_EPSILON;
  float oc[3];
  v_sub(orig, sphere, oc);
  float b = orc_dot(oc, dir);
  float cc = orc_dot(oc, oc) - sphere[3] * sphere[3];
  float disc = b * b - cc;
  if (disc < 0) return 0;
  float sq = sqrtf(disc);
  float t2 = -b - sq;
  if (t2 < eps) t2 = -b + sq;
  if (t2 < eps) return 0;
  if (t2 >= *t) return 0;
  *t = t2;
  return 1;
}

/* ------------------------------------------------------------- hit loops */

typedef struct {
  int kind;       /* 0 none, 1 triangle, 2 sphere */
  int32_t geomID; /* triangle: geometry id */
  int32_t primID; /* triangle: face id; sphere: sphere index */
} orc_hit;

/* main.cpp:176-192 cpp_intersect as called from main.cpp:302-312 intersect():
 * both `u` and `v` reference arguments alias the caller's v (quirk S1). */
static int closest_hit(const orc_scene *s, const float ori[3], const float dir[3], float *t,
                       float *u, float *v, int quirk_s1, orc_hit *hit) {
  hit->kind = 0;
  hit->geomID = -1;
  hit->primID = -1;
  for (int32_t i = 0; i < s->n_geometry; i++) {
    const orc_geometry *g = &s->geometry[i];
    for (int32_t f = 0; f < g->n_faces; f++) {
      const uint32_t *face = &g->face_index[3 * f];
      float uu, vv;
      if (orc_intersect_triangle(ori, dir, &g->vertex[3 * face[0]], &g->vertex[3 * face[1]],
                                 &g->vertex[3 * face[2]], t, &uu, &vv)) {
        if (quirk_s1) {
          *v = vv; /* u = u2 then v = v2 through the same reference; caller's u untouched */
        } else {
          *u = uu;
          *v = vv;
        }
        hit->kind = 1;
        hit->geomID = i;
        hit->primID = f;
      }
    }
  }
  /* extension: spheres come after every triangle in tie order */
  for (int32_t k = 0; k < s->n_spheres; k++) {
    if (orc_intersect_sphere(ori, dir, &s->spheres[4 * k], t)) {
      hit->kind = 2;
      hit->geomID = -1;
      hit->primID = k;
    }
  }
  return hit->kind != 0;
}

/* main.cpp:314-329: first accepted primitive returns; *t keeps that primitive's t2 */
static int occlusion(const orc_scene *s, const float ori[3], const float dir[3], float *t,
                     uint64_t *tests) {
  float u, v;
  for (int32_t i = 0; i < s->n_geometry; i++) {
    const orc_geometry *g = &s->geometry[i];
    for (int32_t f = 0; f < g->n_faces; f++) {
      const uint32_t *face = &g->face_index[3 * f];
      (*tests)++;
      if (orc_intersect_triangle(ori, dir, &g->vertex[3 * face[0]], &g->vertex[3 * face[1]],
                                 &g->vertex[3 * face[2]], t, &u, &v))
        return 1;
    }
  }
  for (int32_t k = 0; k < s->n_spheres; k++) {
    (*tests)++;
    if (orc_intersect_sphere(ori, dir, &s->spheres[4 * k], t)) return 1;
  }
  return 0;
}

/* splitmix64 finaliser over (seed, pixel, light); replaces the reference's
 * std::random_device-seeded mt19937 draw (main.cpp:587-588,743-747), which is not
 * reproducible run to run. */
uint32_t orc_face_hash(uint64_t seed, uint32_t pixel, uint32_t light, uint32_t n_faces) {
  uint64_t z = seed + (((uint64_t)pixel << 32) | (uint64_t)light) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)((z >> 32) % (uint64_t)n_faces);
}

/* ---------------------------------------------------------------- scan_row */

/* main.cpp:698-791, triangle branch (num_triangles == 0) */
static void scan_row(const orc_scene *s, const orc_camera *cam, int32_t W, int32_t H, int32_t h,
                     const orc_options *o, float *row_out /* W*3 */, orc_counters *cnt) {
  const float eps = FLT_EPSILON;
  const int quirk_s1 = (o->quirks & ORC_QUIRK_S1) != 0;
  const int quirk_s3 = (o->quirks & ORC_QUIRK_S3) != 0;
  for (int32_t w = 0; w < W; w++) {
    float *px = &row_out[(int64_t)w * 3];
    px[0] = px[1] = px[2] = 0.f; /* vec3 default ctor zero-fills, main.cpp:557-558 */
    cnt->primary_rays++;

    float is = (float)w / (W - 1); /* :709 */
    float it = (float)h / (H - 1); /* :710 */
    float dir[3];
    orc_camera_get_ray(cam, is, it, dir); /* :713 */
    const float *origin = cam->origin;

    float t = FLT_MAX; /* :715 */
    float u = 0;
    float v = 0;
    orc_hit hit;
    if (!closest_hit(s, origin, dir, &t, &u, &v, quirk_s1, &hit)) continue; /* :722 */
    cnt->hit_pixels++;

    float N[3];
    const orc_material *mat;
    if (hit.kind == 1) {
      const orc_geometry *g = &s->geometry[hit.geomID];
      const uint32_t *face = &g->face_index[3 * hit.primID];
      float e1[3], e2[3], cr[3];
      /* :728-731 normalize(cross(v1 - v0, v2 - v0)) */
      v_sub(&g->vertex[3 * face[1]], &g->vertex[3 * face[0]], e1);
      v_sub(&g->vertex[3 * face[2]], &g->vertex[3 * face[0]], e2);
      orc_cross(e1, e2, cr);
      orc_normalize(cr, N);
      if (g->n_normals != 0) { /* :733-738 */
        const float *N0 = &g->normals[3 * face[0]];
        const float *N1 = &g->normals[3 * face[1]];
        const float *N2 = &g->normals[3 * face[2]];
        float a[3], b[3], c3[3];
        v_scale(N1, u, a);
        v_scale(N2, v, b);
        v_add(a, b, a);
        v_scale(N0, (1 - u - v), c3); /* (1.0f - u) - v */
        v_add(a, c3, a);
        orc_normalize(a, N);
      }
      mat = &g->material; /* :768 */
    } else {
      /* extension: N = normalize((o + d*t) - c) */
      const float *sp = &s->spheres[4 * hit.primID];
      float p[3];
      v_scale(dir, t, p);
      v_add(origin, p, p);
      v_sub(p, sp, p);
      orc_normalize(p, N);
      mat = &s->sphere_materials[s->sphere_material[hit.primID]];
    }

    const float t_hit = t;
    const float nl = (float)s->n_lights; /* float(light_sources.size()) */
    for (int32_t li = 0; li < s->n_lights; li++) { /* :740 */
      const orc_geometry *light = &s->geometry[s->light_sources[li]];
      uint32_t faceID;
      if (o->face_mode == ORC_FACE_FIXED)
        faceID = (uint32_t)o->fixed_face;
      else
        faceID = orc_face_hash(o->seed, (uint32_t)(h * W + w), (uint32_t)li,
                               (uint32_t)light->n_faces);
      /* :748-754 quirk S2: v0 = v1 = v2 = light.vertex[faceID], so
       * P = v0 + ((v1-v0)*r1 + (v2-v0)*r2) = v0 + (+0) for any finite draws r1,r2 */
      const float *lv = &light->vertex[3 * faceID];
      float P[3] = {lv[0] + 0.0f, lv[1] + 0.0f, lv[2] + 0.0f};

      float t_use = quirk_s3 ? t : t_hit;
      float hitp[3], L[3];
      v_scale(dir, (t_use - eps), hitp); /* :757-758 */
      v_add(origin, hitp, hitp);
      v_sub(P, hitp, L);            /* :759 */
      float len = orc_length(L);    /* :761 */
      t = len - eps;                /* :764 */
      orc_normalize(L, L);          /* :766 */

      float c[3], tmp[3];
      v_scale(mat->ka, 0.5f, c); /* :769-770 (ka*0.5f + ke) / float(nl) */
      v_add(c, mat->ke, c);
      v_div(c, nl, c);

      if (o->shadows) {
        cnt->shadow_rays++;
        if (occlusion(s, hitp, L, &t, &cnt->anyhit_tests)) continue; /* :772 */
      }
      float d = orc_dot(N, L); /* :775 */
      if (d <= 0) continue;    /* :777 */

      float Hh[3];
      v_add(N, L, tmp);
      v_scale(tmp, 2.f, tmp);
      orc_normalize(tmp, Hh); /* :780 */

      /* :782-783 c + (kd*d + ks*pow(dot(N,H), Ns)) / float(nl) */
      float spec = powf(orc_dot(N, Hh), mat->Ns);
      float kd_d[3], ks_p[3];
      v_scale(mat->kd, d, kd_d);
      v_scale(mat->ks, spec, ks_p);
      v_add(kd_d, ks_p, tmp);
      v_div(tmp, nl, tmp);
      v_add(c, tmp, c);

      px[0] += c[0]; /* :786-788 */
      px[1] += c[1];
      px[2] += c[2];
    }
  }
}

typedef struct {
  const orc_scene *scene;
  const orc_camera *cam;
  int32_t W, H, tid, n_threads;
  const int32_t *rows; /* image rows to render */
  int32_t n_rows;
  int packed; /* 1: i-th listed row goes to image row i; 0: to image row rows[i] */
  const orc_options *opts;
  float *image;
  orc_counters cnt;
} worker_arg;

static void *worker(void *p) {
  worker_arg *a = (worker_arg *)p;
  /* rows dealt round-robin; each pixel is a pure function of (w,h) */
  for (int32_t i = a->tid; i < a->n_rows; i += a->n_threads) {
    const int32_t h = a->rows[i];
    float *out = a->image + (int64_t)(a->packed ? i : h) * a->W * 3;
    scan_row(a->scene, a->cam, a->W, a->H, h, a->opts, out, &a->cnt);
  }
  return NULL;
}

static void run_rows(const orc_scene *scene, const orc_camera *cam, int32_t W, int32_t H,
                     const int32_t *rows, int32_t n_rows, int packed, const orc_options *opts,
                     float *image, orc_counters *counters, int32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  worker_arg *args = (worker_arg *)calloc((size_t)n_threads, sizeof(worker_arg));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int32_t i = 0; i < n_threads; i++) {
    worker_arg *a = &args[i];
    a->scene = scene;
    a->cam = cam;
    a->W = W;
    a->H = H;
    a->tid = i;
    a->n_threads = n_threads;
    a->rows = rows;
    a->n_rows = n_rows;
    a->packed = packed;
    a->opts = opts;
    a->image = image;
    if (n_threads == 1)
      worker(a);
    else
      pthread_create(&th[i], NULL, worker, a);
  }
  orc_counters total = {0, 0, 0, 0};
  for (int32_t i = 0; i < n_threads; i++) {
    if (n_threads > 1) pthread_join(th[i], NULL);
    total.primary_rays += args[i].cnt.primary_rays;
    total.hit_pixels += args[i].cnt.hit_pixels;
    total.shadow_rays += args[i].cnt.shadow_rays;
    total.anyhit_tests += args[i].cnt.anyhit_tests;
  }
  if (counters) *counters = total;
  free(args);
  free(th);
}

void orc_render(const orc_scene *scene, const orc_camera *cam, int32_t W, int32_t H,
                int32_t row_begin, int32_t row_end, const orc_options *opts, float *image,
                orc_counters *counters, int32_t n_threads) {
  int32_t n = row_end > row_begin ? row_end - row_begin : 0;
  int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int32_t i = 0; i < n; i++) rows[i] = row_end - 1 - i; /* top-down like main.cpp:628 */
  run_rows(scene, cam, W, H, rows, n, 0, opts, image, counters, n_threads);
  free(rows);
}

void orc_render_row_list(const orc_scene *scene, const orc_camera *cam, int32_t W, int32_t H,
                         const int32_t *rows, int32_t n_rows, const orc_options *opts,
                         float *packed_rows, orc_counters *counters, int32_t n_threads) {
  run_rows(scene, cam, W, H, rows, n_rows, 1, opts, packed_rows, counters, n_threads);
}

/* ------------------------------------------------------------ PPM (S12) */

/* main.cpp:676-682 */
void orc_quantise(const float *image, int64_t n_values, uint8_t *out) {
  for (int64_t i = 0; i < n_values; i++) {
    float c = image[i];
    c = (c > 1.f) ? 1.f : c;
    out[i] = (uint8_t)(int)(c * 255);
  }
}

/* main.cpp:661-685 */
int orc_write_ppm(const char *path, const float *image, int32_t W, int32_t H) {
  FILE *f = fopen(path, "w");
  if (!f) return -1;
  fprintf(f, "P3\n%d %d\n255\n", W, H);
  for (int32_t h = H - 1; h >= 0; --h) {
    for (int32_t w = 0; w < W; ++w) {
      const float *px = &image[((int64_t)h * W + w) * 3];
      float r = (px[0] > 1.f) ? 1.f : px[0];
      float g = (px[1] > 1.f) ? 1.f : px[1];
      float b = (px[2] > 1.f) ? 1.f : px[2];
      fprintf(f, "%d %d %d\n", (int)(r * 255), (int)(g * 255), (int)(b * 255));
    }
  }
  return fclose(f) == 0 ? 0 : -1;
}
