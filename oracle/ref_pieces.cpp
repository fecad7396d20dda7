// ref_pieces.cpp -- thin extern "C" doorways onto the REFERENCE's own code.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit #includes the reference's
// headers where they lie (-I/root/reference/src, nothing copied, nothing
// modified, no stand-in headers) and is linked with the reference's
// src/scene/sceneloader.cpp compiled in place.  Output goes to oracle/_ref/ only
// (git-ignored).  It exists so tests can check oracle/rt_oracle.c bit-for-bit
// against the real vec.h / camera.h / ray_triangle.h / sceneloader.cpp.
//
// What is NOT here: scan_row / cpp_intersect / occlusion (src/main.cpp).  main.cpp
// includes the ISPC-generated "trace_ispc.h" unconditionally (main.cpp:25) and the
// `ispc` compiler is not in this image, so that file is unbuildable here and no
// stand-in header is written for it ("Oracle").
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "math/vec.h"
#include "scene/camera.h"
#include "scene/ray_triangle.h"
#include "scene/sceneloader.h"

using tracer::vec3;

static vec3<float> V(const float *p) { return vec3<float>(p[0], p[1], p[2]); }
static void out3(const vec3<float> &v, float *o) {
  o[0] = v.x;
  o[1] = v.y;
  o[2] = v.z;
}

extern "C" {

float ref_dot(const float *a, const float *b) { return tracer::dot(V(a), V(b)); }
void ref_cross(const float *a, const float *b, float *o) { out3(tracer::cross(V(a), V(b)), o); }
void ref_normalize(const float *a, float *o) { out3(tracer::normalize(V(a)), o); }
float ref_length(const float *a) { return tracer::length(V(a)); }
void ref_add(const float *a, const float *b, float *o) { out3(V(a) + V(b), o); }
void ref_sub(const float *a, const float *b, float *o) { out3(V(a) - V(b), o); }
void ref_scale(const float *a, float s, float *o) { out3(V(a) * s, o); }
void ref_div(const float *a, float s, float *o) { out3(V(a) / s, o); }

// camera.h:16-29 -> 12 floats: origin, lower_left_corner, horizontal, vertical
void ref_camera(const float *from, const float *at, const float *vup, float vfov, float aspect,
                float *out12) {
  tracer::camera cam(V(from), V(at), V(vup), vfov, aspect);
  out3(cam.origin, out12);
  out3(cam.lower_left_corner, out12 + 3);
  out3(cam.horizontal, out12 + 6);
  out3(cam.vertical, out12 + 9);
}

// camera.h:31-34
void ref_get_ray(const float *from, const float *at, const float *vup, float vfov, float aspect,
                 float s, float t, float *dir_out) {
  tracer::camera cam(V(from), V(at), V(vup), vfov, aspect);
  tracer::ray r = cam.get_ray(s, t);
  out3(r.dir, dir_out);
}

// ray_triangle.h:7-57
int ref_intersect_triangle(const float *orig, const float *dir, const float *v0, const float *v1,
                           const float *v2, float *t, float *u, float *v) {
  return tracer::intersect_triangle(V(orig), V(dir), V(v0), V(v1), V(v2), *t, *u, *v) ? 1 : 0;
}

// ---- sceneloader.cpp:14-106 model::loadobj, result flattened for ctypes ----
struct ref_scene_dump {
  tracer::scene scene;
  std::string error;
};

void *ref_loadobj(const char *path) {
  ref_scene_dump *d = new ref_scene_dump();
  try {
    d->scene = model::loadobj(path);
  } catch (const std::exception &e) {
    d->error = e.what();
    if (d->error.empty()) d->error = "exception";
  }
  return d;
}
void ref_scene_free(void *p) { delete static_cast<ref_scene_dump *>(p); }
const char *ref_scene_error(void *p) { return static_cast<ref_scene_dump *>(p)->error.c_str(); }
int ref_scene_n_geometry(void *p) {
  return (int)static_cast<ref_scene_dump *>(p)->scene.geometry.size();
}
int ref_scene_n_lights(void *p) {
  return (int)static_cast<ref_scene_dump *>(p)->scene.light_sources.size();
}
int ref_scene_light(void *p, int i) {
  return (int)static_cast<ref_scene_dump *>(p)->scene.light_sources[i];
}
// counts[0..2] = n_vertices, n_normals, n_faces
void ref_geom_counts(void *p, int g, int *counts) {
  auto &G = static_cast<ref_scene_dump *>(p)->scene.geometry[g];
  counts[0] = (int)G.vertex.size();
  counts[1] = (int)G.normals.size();
  counts[2] = (int)G.face_index.size();
}
void ref_geom_copy(void *p, int g, float *vertex, float *normals, uint32_t *faces,
                   float *material13) {
  auto &G = static_cast<ref_scene_dump *>(p)->scene.geometry[g];
  for (size_t i = 0; i < G.vertex.size(); i++) out3(G.vertex[i], vertex + 3 * i);
  for (size_t i = 0; i < G.normals.size(); i++) out3(G.normals[i], normals + 3 * i);
  for (size_t i = 0; i < G.face_index.size(); i++)
    for (unsigned k = 0; k < 3; k++) faces[3 * i + k] = G.face_index[i][k];
  const auto &m = G.object_material;
  out3(m.ka, material13);
  out3(m.kd, material13 + 3);
  out3(m.ks, material13 + 6);
  out3(m.ke, material13 + 9);
  material13[12] = m.Ns;
}

} // extern "C"
