/*
 * rt_oracle_fast.c -- the same restatement as rt_oracle.c, eight pixels at a time.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h): it is the CPU baseline bench.py times
 * (SURVEY.md 8(d) "cpu_fast": the stand-in for the reference's host ISPC path, which
 * cannot be built here), and tests/test_oracle_pins.py checks it bit for bit against
 * rt_oracle.c on every fixture scene.
 *
 * How it stays bit-exact while being vectorised: the eight lanes of a packet are eight
 * independent pixels; every lane performs exactly the scalar sequence of rt_oracle.c
 * (same operations, same order, fp32 / fp64 where the reference has them), so the
 * compiler may run the lanes in one AVX2 register but cannot change any lane's result:
 *   gcc -O3 -mavx2 -ffp-contract=off -fno-fast-math -fno-math-errno -fno-trapping-math
 * (-fno-math-errno lets sqrtf become vsqrtps, -fno-trapping-math lets the compiler turn the
 * rejects into selects; neither changes a value: no reassociation, no fusing).  Primitives are
 * walked in the reference's order (main.cpp:179-186); a lane whose test the scalar code
 * would not run (already occluded) is masked.  Only the reference's own behaviour is
 * implemented (all quirks on).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "rt_oracle.h"

#define P 8 /* lanes per packet */

typedef struct {
  float ox[P], oy[P], oz[P]; /* origin */
  float dx[P], dy[P], dz[P]; /* direction */
} packet;

typedef struct { /* closest-hit state, main.cpp:715-722 */
  float t[P], v[P];
  int32_t kind[P], geom[P], prim[P]; /* kind: 0 none, 1 triangle, 2 sphere */
} hits;

/* ray_triangle.h:7-57 for eight rays against one triangle.  accept[l] = the scalar code
 * returns 1 for lane l; t2/v2 = the values it would store. */
static inline void tri8(const packet *r, const float v0[3], const float e1[3], const float e2[3],
                        const float tbound[P], int accept[P], float t2o[P], float v2o[P]) {
  const float eps = FLT_EPSILON;
  float det[P], un[P], vn[P], tn[P];
  for (int l = 0; l < P; l++) {
    /* :18 pvec = cross(dir, edge2) */
    const float px = r->dy[l] * e2[2] - r->dz[l] * e2[1];
    const float py = r->dz[l] * e2[0] - r->dx[l] * e2[2];
    const float pz = r->dx[l] * e2[1] - r->dy[l] * e2[0];
    /* :21 det = dot(edge1, pvec), vec.h:95-101 order with its leading 0 */
    float s = 0;
    s += e1[0] * px;
    s += e1[1] * py;
    s += e1[2] * pz;
    det[l] = s;
    /* :29 tvec = orig - vert0 */
    const float tx = r->ox[l] - v0[0], ty = r->oy[l] - v0[1], tz = r->oz[l] - v0[2];
    s = 0; /* :32 numerator */
    s += tx * px;
    s += ty * py;
    s += tz * pz;
    un[l] = s;
    /* :37 qvec = cross(tvec, edge1) */
    const float qx = ty * e1[2] - tz * e1[1];
    const float qy = tz * e1[0] - tx * e1[2];
    const float qz = tx * e1[1] - ty * e1[0];
    s = 0; /* :40 numerator */
    s += r->dx[l] * qx;
    s += r->dy[l] * qy;
    s += r->dz[l] * qz;
    vn[l] = s;
    s = 0; /* :45 numerator */
    s += e2[0] * qx;
    s += e2[1] * qy;
    s += e2[2] * qz;
    tn[l] = s;
  }
  for (int l = 0; l < P; l++) {
    const double d = det[l];                         /* :21 widened */
    const int ok0 = !((d > -eps) & (d < eps));       /* :23-25 (bitwise: no branches, vectorises) */
    const double inv = 1.0f / d;                     /* :26 */
    const float u2 = (float)((double)un[l] * inv);   /* :32 */
    const int ok1 = !((u2 < eps) | (u2 > 1.0f));     /* :33 */
    const float v2 = (float)((double)vn[l] * inv);   /* :40 */
    const int ok2 = !((v2 < eps) | (u2 + v2 > 1.0f)); /* :41 */
    const float t2 = (float)((double)tn[l] * inv);   /* :45 */
    const int ok3 = !(t2 < eps);                     /* :46 */
    const int ok4 = !(t2 >= tbound[l]);              /* :49 */
    accept[l] = ok0 & ok1 & ok2 & ok3 & ok4;
    t2o[l] = t2;
    v2o[l] = v2;
  }
}

/* sphere extension, rt_oracle.c orc_intersect_sphere, eight rays */
static inline void sph8(const packet *r, const float sp[4], const float tbound[P], int accept[P],
                        float t2o[P]) {
  const float eps = FLT_EPSILON;
  const float r2 = sp[3] * sp[3];
  for (int l = 0; l < P; l++) {
    const float cx = r->ox[l] - sp[0], cy = r->oy[l] - sp[1], cz = r->oz[l] - sp[2];
    float b = 0;
    b += cx * r->dx[l];
    b += cy * r->dy[l];
    b += cz * r->dz[l];
    float cc = 0;
    cc += cx * cx;
    cc += cy * cy;
    cc += cz * cz;
    cc = cc - r2;
    const float disc = b * b - cc;
    const float sq = sqrtf(disc); /* NaN for disc < 0: that lane is rejected below */
    float t2 = -b - sq;
    const float t2b = -b + sq;
    t2 = (t2 < eps) ? t2b : t2;
    accept[l] = !(disc < 0) & !(t2 < eps) & !(t2 >= tbound[l]);
    t2o[l] = t2;
  }
}

/* main.cpp:176-192 (+ spheres after every triangle) */
static void closest8(const orc_scene *s, const packet *r, const int valid[P], hits *h) {
  for (int l = 0; l < P; l++) {
    h->t[l] = FLT_MAX; /* main.cpp:715 */
    h->v[l] = 0;
    h->kind[l] = 0;
    h->geom[l] = -1;
    h->prim[l] = -1;
  }
  int acc[P];
  float t2[P], v2[P];
  for (int32_t i = 0; i < s->n_geometry; i++) {
    const orc_geometry *g = &s->geometry[i];
    for (int32_t f = 0; f < g->n_faces; f++) {
      const uint32_t *face = &g->face_index[3 * f];
      const float *a = &g->vertex[3 * face[0]], *b = &g->vertex[3 * face[1]],
                  *c = &g->vertex[3 * face[2]];
      const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}; /* ray_triangle.h:14 */
      const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}; /* :15 */
      tri8(r, a, e1, e2, h->t, acc, t2, v2);
      for (int l = 0; l < P; l++)
        if (acc[l] & valid[l]) {
          h->t[l] = t2[l];
          h->v[l] = v2[l]; /* quirk S1: only v survives */
          h->kind[l] = 1;
          h->geom[l] = i;
          h->prim[l] = f;
        }
    }
  }
  for (int32_t k = 0; k < s->n_spheres; k++) {
    sph8(r, &s->spheres[4 * k], h->t, acc, t2);
    for (int l = 0; l < P; l++)
      if (acc[l] & valid[l]) {
        h->t[l] = t2[l];
        h->kind[l] = 2;
        h->geom[l] = -1;
        h->prim[l] = k;
      }
  }
}

/* main.cpp:314-329: first accepted primitive ends a lane's scan; *t keeps its t2.
 * looking[l] in: lane has a shadow ray; out occluded[l].  tests[l] = primitive tests the
 * scalar code would have executed for that lane. */
static void occlusion8(const orc_scene *s, const packet *r, float t[P], const int looking_in[P],
                       int occluded[P], uint64_t *tests_total) {
  int looking[P], acc[P], n_look = 0;
  float t2[P], v2[P];
  uint64_t tests[P];
  for (int l = 0; l < P; l++) {
    looking[l] = looking_in[l];
    occluded[l] = 0;
    tests[l] = 0;
    n_look += looking[l];
  }
  for (int32_t i = 0; i < s->n_geometry && n_look; i++) {
    const orc_geometry *g = &s->geometry[i];
    for (int32_t f = 0; f < g->n_faces && n_look; f++) {
      const uint32_t *face = &g->face_index[3 * f];
      const float *a = &g->vertex[3 * face[0]], *b = &g->vertex[3 * face[1]],
                  *c = &g->vertex[3 * face[2]];
      const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
      const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      tri8(r, a, e1, e2, t, acc, t2, v2);
      for (int l = 0; l < P; l++)
        if (looking[l]) {
          tests[l]++;
          if (acc[l]) {
            t[l] = t2[l];
            occluded[l] = 1;
            looking[l] = 0;
            n_look--;
          }
        }
    }
  }
  for (int32_t k = 0; k < s->n_spheres && n_look; k++) {
    sph8(r, &s->spheres[4 * k], t, acc, t2);
    for (int l = 0; l < P; l++)
      if (looking[l]) {
        tests[l]++;
        if (acc[l]) {
          t[l] = t2[l];
          occluded[l] = 1;
          looking[l] = 0;
          n_look--;
        }
      }
  }
  for (int l = 0; l < P; l++) *tests_total += tests[l];
}

static float dot3(const float a[3], const float b[3]) { /* vec.h:95-101 */
  float sum = 0;
  for (int i = 0; i < 3; i++) sum += (a[i] * b[i]);
  return sum;
}

/* main.cpp:698-791 for one row, eight pixels at a time */
static void scan_row8(const orc_scene *s, const orc_camera *cam, int32_t W, int32_t H, int32_t h,
                      const orc_options *o, float *row_out, orc_counters *cnt) {
  const float eps = FLT_EPSILON;
  const float nl = (float)s->n_lights;
  for (int32_t w0 = 0; w0 < W; w0 += P) {
    packet r;
    int valid[P];
    float dir[P][3];
    for (int l = 0; l < P; l++) {
      const int32_t w = (w0 + l < W) ? w0 + l : W - 1;
      valid[l] = (w0 + l < W);
      const float is = (float)w / (W - 1); /* :709 */
      const float it = (float)h / (H - 1); /* :710 */
      orc_camera_get_ray(cam, is, it, dir[l]);
      r.ox[l] = cam->origin[0]; r.oy[l] = cam->origin[1]; r.oz[l] = cam->origin[2];
      r.dx[l] = dir[l][0]; r.dy[l] = dir[l][1]; r.dz[l] = dir[l][2];
    }
    hits hh;
    closest8(s, &r, valid, &hh);

    float N[P][3], px[P][3], t[P];
    const orc_material *mat[P];
    int has[P];
    for (int l = 0; l < P; l++) {
      px[l][0] = px[l][1] = px[l][2] = 0.f;
      has[l] = valid[l] && hh.kind[l] != 0;
      t[l] = hh.t[l];
      mat[l] = NULL;
      if (valid[l]) cnt->primary_rays++;
      if (!has[l]) continue;
      cnt->hit_pixels++;
      if (hh.kind[l] == 1) { /* :728-738 */
        const orc_geometry *g = &s->geometry[hh.geom[l]];
        const uint32_t *face = &g->face_index[3 * hh.prim[l]];
        const float *a = &g->vertex[3 * face[0]], *b = &g->vertex[3 * face[1]],
                    *c = &g->vertex[3 * face[2]];
        const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
        const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        float cr[3];
        orc_cross(e1, e2, cr);
        orc_normalize(cr, N[l]);
        if (g->n_normals != 0) {
          const float *N0 = &g->normals[3 * face[0]], *N1 = &g->normals[3 * face[1]],
                      *N2 = &g->normals[3 * face[2]];
          const float u = 0, v = hh.v[l]; /* quirk S1 */
          float q[3];
          for (int k = 0; k < 3; k++) q[k] = (N1[k] * u + N2[k] * v) + N0[k] * ((1 - u) - v);
          orc_normalize(q, N[l]);
        }
        mat[l] = &g->material;
      } else { /* extension: N = normalize((o + d*t) - c) */
        const float *sp = &s->spheres[4 * hh.prim[l]];
        float q[3];
        for (int k = 0; k < 3; k++) q[k] = (cam->origin[k] + dir[l][k] * hh.t[l]) - sp[k];
        orc_normalize(q, N[l]);
        mat[l] = &s->sphere_materials[s->sphere_material[hh.prim[l]]];
      }
    }

    for (int32_t li = 0; li < s->n_lights; li++) { /* :740 */
      const orc_geometry *light = &s->geometry[s->light_sources[li]];
      packet sr;
      float L[P][3];
      int occ[P];
      memset(&sr, 0, sizeof(sr));
      for (int l = 0; l < P; l++) {
        occ[l] = 0;
        if (!has[l]) continue;
        uint32_t faceID;
        if (o->face_mode == ORC_FACE_FIXED)
          faceID = (uint32_t)o->fixed_face;
        else
          faceID = orc_face_hash(o->seed, (uint32_t)(h * W + (w0 + l)), (uint32_t)li,
                                 (uint32_t)light->n_faces);
        const float *lv = &light->vertex[3 * faceID]; /* quirk S2 */
        const float Pp[3] = {lv[0] + 0.0f, lv[1] + 0.0f, lv[2] + 0.0f};
        float hp[3];
        for (int k = 0; k < 3; k++) hp[k] = cam->origin[k] + dir[l][k] * (t[l] - eps); /* :757-758 */
        for (int k = 0; k < 3; k++) L[l][k] = Pp[k] - hp[k];                          /* :759 */
        const float len = sqrtf(dot3(L[l], L[l]));                                    /* :761 */
        t[l] = len - eps;                                                             /* :764 */
        orc_normalize(L[l], L[l]);                                                    /* :766 */
        sr.ox[l] = hp[0]; sr.oy[l] = hp[1]; sr.oz[l] = hp[2];
        sr.dx[l] = L[l][0]; sr.dy[l] = L[l][1]; sr.dz[l] = L[l][2];
      }
      if (o->shadows) {
        for (int l = 0; l < P; l++) cnt->shadow_rays += has[l];
        occlusion8(s, &sr, t, has, occ, &cnt->anyhit_tests); /* :772 */
      }
      for (int l = 0; l < P; l++) {
        if (!has[l] || occ[l]) continue;
        const float d = dot3(N[l], L[l]); /* :775 */
        if (d <= 0) continue;             /* :777 */
        const orc_material *m = mat[l];
        float c[3], Hh[3], tmp[3];
        for (int k = 0; k < 3; k++) c[k] = (m->ka[k] * 0.5f + m->ke[k]) / nl; /* :769-770 */
        for (int k = 0; k < 3; k++) tmp[k] = (N[l][k] + L[l][k]) * 2.f;
        orc_normalize(tmp, Hh); /* :780 */
        const float spec = powf(dot3(N[l], Hh), m->Ns);
        for (int k = 0; k < 3; k++) c[k] = c[k] + (m->kd[k] * d + m->ks[k] * spec) / nl; /* :782-783 */
        px[l][0] += c[0]; /* :786-788 */
        px[l][1] += c[1];
        px[l][2] += c[2];
      }
    }
    for (int l = 0; l < P; l++)
      if (valid[l]) memcpy(&row_out[(int64_t)(w0 + l) * 3], px[l], 12);
  }
}

typedef struct {
  const orc_scene *scene;
  const orc_camera *cam;
  int32_t W, H, tid, n_threads, n_rows;
  const int32_t *rows;
  const orc_options *opts;
  float *image;
  orc_counters cnt;
} worker_arg;

static void *worker(void *p) {
  worker_arg *a = (worker_arg *)p;
  for (int32_t i = a->tid; i < a->n_rows; i += a->n_threads)
    scan_row8(a->scene, a->cam, a->W, a->H, a->rows[i], a->opts,
              a->image + (int64_t)i * a->W * 3, &a->cnt);
  return NULL;
}

/* same contract as orc_render_row_list: listed row i goes to image row i */
void orc_fast_render_row_list(const orc_scene *scene, const orc_camera *cam, int32_t W, int32_t H,
                              const int32_t *rows, int32_t n_rows, const orc_options *opts,
                              float *image, orc_counters *counters, int32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  worker_arg *args = (worker_arg *)calloc((size_t)n_threads, sizeof(worker_arg));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int32_t i = 0; i < n_threads; i++) {
    args[i] = (worker_arg){scene, cam, W, H, i, n_threads, n_rows, rows, opts, image, {0, 0, 0, 0}};
    if (i) pthread_create(&th[i], NULL, worker, &args[i]);
  }
  worker(&args[0]);
  for (int32_t i = 1; i < n_threads; i++) pthread_join(th[i], NULL);
  if (counters) {
    memset(counters, 0, sizeof(*counters));
    for (int32_t i = 0; i < n_threads; i++) {
      counters->primary_rays += args[i].cnt.primary_rays;
      counters->hit_pixels += args[i].cnt.hit_pixels;
      counters->shadow_rays += args[i].cnt.shadow_rays;
      counters->anyhit_tests += args[i].cnt.anyhit_tests;
    }
  }
  free(args);
  free(th);
}
