// host_core.cpp -- host half of the C ABI (include/esctp1_rt.h): scene building and
// introspection, camera, ISPC-compatible flattening, PPM output.  No device code here.
// Compiled with -ffp-contract=off: the camera constructor's fp32 arithmetic is part of the
// image (camera.h:16-29) and must match the reference's strict evaluation.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "scene.h"

namespace esc {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

void material_from_floats(const float m[ESC_MATERIAL_FLOATS], Material &out) {
  std::memcpy(out.ka, m + 0, 12);
  std::memcpy(out.kd, m + 3, 12);
  std::memcpy(out.ks, m + 6, 12);
  std::memcpy(out.ke, m + 9, 12);
  out.Ns = m[12];
  // sceneloader.cpp:63-64: lightsource = dot(ke,ke) > 0 (vec.h:95-101 order)
  float sum = 0;
  for (int i = 0; i < 3; i++) sum += out.ke[i] * out.ke[i];
  out.lightsource = sum > 0;
}

void material_to_floats(const Material &m, float out[ESC_MATERIAL_FLOATS]) {
  std::memcpy(out + 0, m.ka, 12);
  std::memcpy(out + 3, m.kd, 12);
  std::memcpy(out + 6, m.ks, 12);
  std::memcpy(out + 9, m.ke, 12);
  out[12] = m.Ns;
}

// vec.h helpers on plain arrays, reference evaluation order
static inline float dot3(const float *a, const float *b) {
  float sum = 0;
  for (int i = 0; i < 3; i++) sum += a[i] * b[i];
  return sum;
}
static inline void cross3(const float *a, const float *b, float *o) {
  float x = a[1] * b[2] - a[2] * b[1];
  float y = a[2] * b[0] - a[0] * b[2];
  float z = a[0] * b[1] - a[1] * b[0];
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
static inline void normalize3(const float *a, float *o) {
  float l = std::sqrt(dot3(a, a));
  o[0] = a[0] / l;
  o[1] = a[1] / l;
  o[2] = a[2] / l;
}

} // namespace esc

using esc::set_error;

extern "C" {

const char *esc_last_error(void) { return esc::g_last_error.c_str(); }
const char *esc_version(void) { return "esctp1raytracer_amd 0.1 gfx950 hip"; }

// ------------------------------------------------------------------ scene
esc_scene *esc_scene_new(void) { return new (std::nothrow) esc_scene(); }
void esc_scene_free(esc_scene *scene) { delete scene; }

int esc_scene_add_geometry(esc_scene *scene, const float *vertex, int32_t n_vertices,
                           const float *normals, int32_t n_normals, const uint32_t *face_index,
                           int32_t n_faces, const float material[ESC_MATERIAL_FLOATS]) {
  if (!scene || !vertex || !face_index || !material || n_vertices <= 0 || n_faces <= 0 ||
      n_normals < 0 || (n_normals > 0 && !normals)) {
    set_error("esc_scene_add_geometry: bad argument");
    return ESC_ERR_INVALID;
  }
  for (int64_t i = 0; i < (int64_t)n_faces * 3; i++) {
    if (face_index[i] >= (uint32_t)n_vertices) {
      set_error("esc_scene_add_geometry: face index out of range");
      return ESC_ERR_INVALID;
    }
    // main.cpp:734-736 reads normals[face[k]] whenever normals is non-empty
    if (n_normals > 0 && face_index[i] >= (uint32_t)n_normals) {
      set_error("esc_scene_add_geometry: face index beyond the normals array");
      return ESC_ERR_INVALID;
    }
  }
  // the kernels' loops and the acceleration structure assume finite coordinates
  for (int64_t i = 0; i < (int64_t)n_vertices * 3; i++)
    if (!std::isfinite(vertex[i])) {
      set_error("esc_scene_add_geometry: non-finite vertex coordinate");
      return ESC_ERR_INVALID;
    }
  esc::Geometry g;
  g.vertex.assign(vertex, vertex + (size_t)n_vertices * 3);
  if (n_normals > 0) g.normals.assign(normals, normals + (size_t)n_normals * 3);
  g.face_index.assign(face_index, face_index + (size_t)n_faces * 3);
  esc::material_from_floats(material, g.object_material);
  scene->geometry.push_back(std::move(g));
  const int id = (int)scene->geometry.size() - 1;
  if (scene->geometry.back().object_material.lightsource) // sceneloader.cpp:102-104
    scene->light_sources.push_back((size_t)id);
  return id;
}

int esc_scene_add_spheres(esc_scene *scene, const float *spheres_xyzr, const float *materials,
                          int32_t n_spheres) {
  if (!scene || !spheres_xyzr || !materials || n_spheres < 0) {
    set_error("esc_scene_add_spheres: bad argument");
    return ESC_ERR_INVALID;
  }
  for (int64_t i = 0; i < (int64_t)n_spheres * 4; i++)
    if (!std::isfinite(spheres_xyzr[i])) {
      set_error("esc_scene_add_spheres: non-finite centre or radius");
      return ESC_ERR_INVALID;
    }
  for (int32_t i = 0; i < n_spheres; i++) {
    esc::Sphere s{spheres_xyzr[4 * i], spheres_xyzr[4 * i + 1], spheres_xyzr[4 * i + 2],
                  spheres_xyzr[4 * i + 3]};
    esc::Material m;
    esc::material_from_floats(materials + (size_t)i * ESC_MATERIAL_FLOATS, m);
    scene->spheres.push_back(s);
    scene->sphere_materials.push_back(m);
  }
  return ESC_OK;
}

int esc_scene_load_obj(esc_scene *scene, const char *obj_path) {
  if (!scene || !obj_path) {
    set_error("esc_scene_load_obj: bad argument");
    return ESC_ERR_INVALID;
  }
  return esc::load_obj(*scene, obj_path);
}

int esc_scene_synthetic(esc_scene *scene, const char *config, int32_t n_override) {
  if (!scene || !config) {
    set_error("esc_scene_synthetic: bad argument");
    return ESC_ERR_INVALID;
  }
  return esc::make_synthetic(*scene, config, n_override);
}

void esc_synthetic_view(float eye[3], float look[3]) {
  eye[0] = 0.f; eye[1] = 3.f; eye[2] = 6.f;
  look[0] = 0.f; look[1] = 2.f; look[2] = -8.f;
}

int esc_scene_get_info(const esc_scene *scene, esc_scene_info *info) {
  if (!scene || !info) {
    set_error("esc_scene_get_info: bad argument");
    return ESC_ERR_INVALID;
  }
  info->n_geometry = (int32_t)scene->geometry.size();
  info->n_lights = (int32_t)scene->light_sources.size();
  info->n_triangles = (int32_t)scene->n_triangles();
  info->n_spheres = (int32_t)scene->spheres.size();
  return ESC_OK;
}

int esc_scene_geometry_counts(const esc_scene *scene, int32_t geom, int32_t counts[3]) {
  if (!scene || geom < 0 || (size_t)geom >= scene->geometry.size() || !counts) {
    set_error("esc_scene_geometry_counts: bad argument");
    return ESC_ERR_INVALID;
  }
  const auto &g = scene->geometry[geom];
  counts[0] = (int32_t)g.n_vertices();
  counts[1] = (int32_t)g.n_normals();
  counts[2] = (int32_t)g.n_faces();
  return ESC_OK;
}

int esc_scene_geometry_copy(const esc_scene *scene, int32_t geom, float *vertex, float *normals,
                            uint32_t *face_index, float material[ESC_MATERIAL_FLOATS]) {
  if (!scene || geom < 0 || (size_t)geom >= scene->geometry.size()) {
    set_error("esc_scene_geometry_copy: bad argument");
    return ESC_ERR_INVALID;
  }
  const auto &g = scene->geometry[geom];
  if (vertex) std::memcpy(vertex, g.vertex.data(), g.vertex.size() * 4);
  if (normals && !g.normals.empty()) std::memcpy(normals, g.normals.data(), g.normals.size() * 4);
  if (face_index) std::memcpy(face_index, g.face_index.data(), g.face_index.size() * 4);
  if (material) esc::material_to_floats(g.object_material, material);
  return ESC_OK;
}

int esc_scene_light_sources(const esc_scene *scene, int32_t *geom_ids) {
  if (!scene || !geom_ids) {
    set_error("esc_scene_light_sources: bad argument");
    return ESC_ERR_INVALID;
  }
  for (size_t i = 0; i < scene->light_sources.size(); i++)
    geom_ids[i] = (int32_t)scene->light_sources[i];
  return ESC_OK;
}

int esc_scene_spheres_copy(const esc_scene *scene, float *spheres_xyzr, float *materials) {
  if (!scene) {
    set_error("esc_scene_spheres_copy: bad argument");
    return ESC_ERR_INVALID;
  }
  for (size_t i = 0; i < scene->spheres.size(); i++) {
    if (spheres_xyzr) {
      spheres_xyzr[4 * i + 0] = scene->spheres[i].cx;
      spheres_xyzr[4 * i + 1] = scene->spheres[i].cy;
      spheres_xyzr[4 * i + 2] = scene->spheres[i].cz;
      spheres_xyzr[4 * i + 3] = scene->spheres[i].r;
    }
    if (materials)
      esc::material_to_floats(scene->sphere_materials[i], materials + i * ESC_MATERIAL_FLOATS);
  }
  return ESC_OK;
}

// ----------------------------------------------------------------- camera
// camera.h:16-29, evaluated exactly as written there
void esc_camera_init(esc_camera *cam, const float lookfrom[3], const float lookat[3],
                     const float vup[3], float vfov, float aspect) {
  float theta = (float)((double)vfov * M_PI / 180); // :19 double product, narrowed
  float half_height = std::tan(theta / 2);          // :20 float overload
  float half_width = aspect * half_height;          // :21
  float w[3], u[3], v[3], d[3];
  for (int i = 0; i < 3; i++) cam->origin[i] = lookfrom[i]; // :22
  for (int i = 0; i < 3; i++) d[i] = lookfrom[i] - lookat[i];
  esc::normalize3(d, w); // :23
  esc::cross3(vup, w, d);
  esc::normalize3(d, u); // :24
  esc::cross3(w, u, v);  // :25
  for (int i = 0; i < 3; i++) {
    // :26 ((origin - u*hw) - v*hh) - w
    cam->lower_left_corner[i] = ((cam->origin[i] - u[i] * half_width) - v[i] * half_height) - w[i];
    cam->horizontal[i] = (u[i] * 2.f) * half_width; // :27
    cam->vertical[i] = (v[i] * 2.f) * half_height;  // :28
  }
}

// ------------------------------------------------- ISPC-compatible flatten
} // extern "C"

struct esc_flat_scene { // FlatScene, flatten_iscp.h:9-13
  std::vector<ispc_triangle> triangles;
  std::vector<ispc_triangle> light_faces;
  std::vector<ispc_light> lights;
  std::vector<std::unique_ptr<std::vector<int32_t>>> light_face_indexes; // keeps I4 from dangling
};

extern "C" {

int esc_flatten_ispc(const esc_scene *scene, int32_t sort_by_centroid_x, esc_flat_scene **out) {
  if (!scene || !out) {
    set_error("esc_flatten_ispc: bad argument");
    return ESC_ERR_INVALID;
  }
  if (!scene->spheres.empty()) {
    set_error("esc_flatten_ispc: ispc_triangle has no sphere form; use esc_upload_scene");
    return ESC_ERR_INVALID;
  }
  auto flat = std::make_unique<esc_flat_scene>();
  for (size_t geom_id = 0; geom_id < scene->geometry.size(); geom_id++) { // flatten_iscp.cpp:37
    const esc::Geometry &geom = scene->geometry[geom_id];
    const esc::Material &mat = geom.object_material;
    const int has_normals = !geom.normals.empty(); // :46
    const int is_light = std::find(scene->light_sources.begin(), scene->light_sources.end(),
                                   geom_id) != scene->light_sources.end(); // :51-54
    auto idx = std::make_unique<std::vector<int32_t>>();
    for (size_t f = 0; f < geom.n_faces(); f++) { // :56
      ispc_triangle t;
      std::memset(&t, 0, sizeof(t)); // the reference leaves normals uninitialised when absent
      t.geom_id = (int32_t)geom_id;
      t.prim_id = (int32_t)f;
      t.has_normals = has_normals;
      t.is_light = is_light;
      std::memcpy(t.ka, mat.ka, 12);
      std::memcpy(t.kd, mat.kd, 12);
      std::memcpy(t.ks, mat.ks, 12);
      std::memcpy(t.ke, mat.ke, 12);
      t.Ns = mat.Ns;
      for (int v = 0; v < 3; v++) { // :76-86
        const uint32_t corner = geom.face_index[3 * f + v];
        std::memcpy(t.vertices[v], &geom.vertex[3 * corner], 12);
        if (has_normals) std::memcpy(t.normals[v], &geom.normals[3 * corner], 12);
      }
      flat->triangles.push_back(t);
      if (is_light) { // :90-97
        flat->light_faces.push_back(t);
        idx->push_back((int32_t)flat->light_faces.size() - 1);
      }
    }
    if (is_light) { // :100-106
      ispc_light l;
      l.geom_id = (int32_t)geom_id;
      l.light_faces = idx->data();
      l.num_light_faces = (int32_t)idx->size();
      flat->lights.push_back(l);
      flat->light_face_indexes.push_back(std::move(idx));
    }
  }
  if (sort_by_centroid_x) { // :110 with the comparator of :14-21
    std::sort(flat->triangles.begin(), flat->triangles.end(),
              [](const ispc_triangle &a, const ispc_triangle &b) {
                float ca = (a.vertices[0][0] + a.vertices[1][0] + a.vertices[2][0]) / 3;
                float cb = (b.vertices[0][0] + b.vertices[1][0] + b.vertices[2][0]) / 3;
                return ca < cb;
              });
  }
  *out = flat.release();
  return ESC_OK;
}

void esc_flat_free(esc_flat_scene *flat) { delete flat; }
ispc_triangle *esc_flat_triangles(esc_flat_scene *flat, int32_t *n) {
  if (n) *n = (int32_t)flat->triangles.size();
  return flat->triangles.data();
}
ispc_triangle *esc_flat_light_triangles(esc_flat_scene *flat, int32_t *n) {
  if (n) *n = (int32_t)flat->light_faces.size();
  return flat->light_faces.data();
}
ispc_light *esc_flat_lights(esc_flat_scene *flat, int32_t *n) {
  if (n) *n = (int32_t)flat->lights.size();
  return flat->lights.data();
}

// flatten_iscp.cpp:117-128
void esc_new_ispc_cam(ispc_cam *cam, const float lookfrom[3], const float lookat[3],
                      const float vup[3], float vfov, float aspect) {
  for (int i = 0; i < 3; i++) {
    cam->lookfrom[i] = lookfrom[i];
    cam->lookat[i] = lookat[i];
    cam->vup[i] = vup[i];
  }
  cam->vfov = vfov;
  cam->aspect = aspect;
}

// -------------------------------------------------------------------- PPM
// main.cpp:676-682
void esc_quantise(const float *image, int64_t n_values, uint8_t *out) {
  for (int64_t i = 0; i < n_values; i++) {
    float c = image[i];
    c = (c > 1.f) ? 1.f : c;
    out[i] = (uint8_t)(int)(c * 255);
  }
}

// main.cpp:661-685: one "r g b\n" per pixel, rows top-down.  The text is assembled in a
// large buffer with a 0..255 -> ASCII table (a 4K frame is ~95 MB of text; the reference
// goes through std::ofstream::operator<<(int) three times per pixel).
int esc_write_ppm_u8(const char *path, const uint8_t *rgb8, int32_t W, int32_t H) {
  if (!path || !rgb8 || W <= 0 || H <= 0) {
    set_error("esc_write_ppm_u8: bad argument");
    return ESC_ERR_INVALID;
  }
  FILE *f = std::fopen(path, "wb");
  if (!f) {
    set_error(std::string("esc_write_ppm: cannot open ") + path);
    return ESC_ERR_IO;
  }
  char lut[256][4];
  int lut_len[256];
  for (int i = 0; i < 256; i++) lut_len[i] = std::snprintf(lut[i], 4, "%d", i);
  std::fprintf(f, "P3\n%d %d\n255\n", W, H);
  std::vector<char> line((size_t)W * 12);
  bool ok = true;
  for (int32_t h = H - 1; h >= 0 && ok; --h) {
    char *p = line.data();
    const uint8_t *row = rgb8 + (size_t)h * W * 3;
    for (int32_t w = 0; w < W; ++w) {
      for (int c = 0; c < 3; c++) {
        const uint8_t q = row[3 * w + c];
        std::memcpy(p, lut[q], (size_t)lut_len[q]);
        p += lut_len[q];
        *p++ = (c == 2) ? '\n' : ' ';
      }
    }
    ok = std::fwrite(line.data(), 1, (size_t)(p - line.data()), f) == (size_t)(p - line.data());
  }
  if (std::fclose(f) != 0) ok = false;
  if (!ok) {
    set_error(std::string("esc_write_ppm: write failed for ") + path);
    return ESC_ERR_IO;
  }
  return ESC_OK;
}

int esc_write_ppm(const char *path, const float *image, int32_t W, int32_t H) {
  if (!path || !image || W <= 0 || H <= 0) {
    set_error("esc_write_ppm: bad argument");
    return ESC_ERR_INVALID;
  }
  std::vector<uint8_t> q((size_t)W * H * 3);
  esc_quantise(image, (int64_t)W * H * 3, q.data());
  return esc_write_ppm_u8(path, q.data(), W, H);
}

} // extern "C"
