// scene.h -- host-side scene model of the MI355X renderer.
//
// Mirrors the load-time API the reference keeps (north_star: "src/scene, src/math and
// src/models stay as the load-time API"): tracer::scene / Geometry / Material of
// /root/reference/src/scene/scene.h:8-44, restated as plain packed arrays (12-byte float3,
// not the reference's 16-byte aligned vec3) because everything here is headed for flat HBM
// tables.  The sphere list is this build's extension.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "esctp1_rt.h"

namespace esc {

struct Material { // scene.h:11-18
  float ka[3] = {0, 0, 0};
  float kd[3] = {0, 0, 0};
  float ks[3] = {0, 0, 0};
  float ke[3] = {0, 0, 0};
  float Ns = 0.f;
  bool lightsource = false;
};

struct Geometry {               // scene.h:20-31
  std::vector<float> vertex;    // xyz per vertex
  std::vector<float> normals;   // xyz per vertex, may be empty or shorter (sceneloader.cpp:84-89)
  std::vector<uint32_t> face_index; // 3 per face
  Material object_material;
  std::string name;
  size_t n_vertices() const { return vertex.size() / 3; }
  size_t n_normals() const { return normals.size() / 3; }
  size_t n_faces() const { return face_index.size() / 3; }
};

struct Sphere {
  float cx, cy, cz, r;
};

} // namespace esc

// the opaque handle of include/esctp1_rt.h
struct esc_scene {
  std::vector<esc::Geometry> geometry;      // scene.h:34
  std::vector<size_t> light_sources;        // scene.h:35
  std::vector<esc::Sphere> spheres;         // extension
  std::vector<esc::Material> sphere_materials; // one per sphere
  size_t n_triangles() const {
    size_t n = 0;
    for (const auto &g : geometry) n += g.n_faces();
    return n;
  }
};

namespace esc {

void set_error(const std::string &msg);
void material_from_floats(const float m[ESC_MATERIAL_FLOATS], Material &out);
void material_to_floats(const Material &m, float out[ESC_MATERIAL_FLOATS]);

// sceneloader.cpp:14-106 restated over our own OBJ/MTL reader (obj_loader.cpp)
int load_obj(esc_scene &scene, const std::string &path);
// SURVEY.md 8(d) synthetic configs (synth.cpp)
int make_synthetic(esc_scene &scene, const std::string &config, int n_override);

} // namespace esc
