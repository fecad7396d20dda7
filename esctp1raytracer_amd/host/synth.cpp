// synth.cpp -- the synthetic scenes BASELINE.json's configs are quoted on.
//
// The reference ships no scene of these shapes (its largest mesh has 7,088 triangles and it
// has no sphere primitive), so the workloads are generated.  All
// constants are frozen here and echoed in BASELINE.md / DESIGN.md:
//   generator   splitmix64, u = (x >> 40) * 2^-24, value = lo + (hi - lo) * u   (fp32)
//   camera      eye (0,3,6) -> look (0,2,-8), vup (0,1,0), vfov 60   (esc_synthetic_view)
//   floor       2 triangles, x in [-12,12], z in [-24,4], y = 0, ka = kd = (0.725,0.71,0.68)
//   light       ONE triangle at y = 12 facing down, ke = (17,12,4), ka = kd = 0.78
//               (material values of CornellBox-Original.mtl's light); a single face keeps the
//               reference's light sampling deterministic
//   all ks = 0, Ns = 10
//   c2  seed 0xC2, 100   spheres, r in [0.2,0.8]
//   c3  seed 0xC3, 1000  spheres, r in [0.1,0.4]
//   c4  seed 0xC4, 10000 spheres, r in [0.05,0.2]
//       centres x in [-8,8], y in [0.5,5], z in [-20,-2]; ka = kd in [0.2,0.9]^3;
//       draw order per sphere: cx cy cz r kr kg kb
//   c5  224 x 224 quads = 100,352 triangles, y = 0.4*sin(0.9x)*cos(0.7z) over the floor
//       rectangle (replaces the floor), same light
#include <cmath>
#include <cstring>
#include <string>

#include "scene.h"

namespace esc {

namespace {

struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  float uniform(float lo, float hi) {
    const float u = (float)(next() >> 40) * (1.0f / 16777216.0f);
    return lo + (hi - lo) * u;
  }
};

Material matte(float r, float g, float b) {
  Material m;
  m.ka[0] = m.kd[0] = r;
  m.ka[1] = m.kd[1] = g;
  m.ka[2] = m.kd[2] = b;
  m.Ns = 10.f;
  return m;
}

void add_triangle_geometry(esc_scene &scene, const std::vector<float> &tris /* 9 per face */,
                           const Material &m, const char *name) {
  Geometry g;
  g.name = name;
  g.vertex = tris; // already de-indexed, 3 vertices per face (sceneloader.cpp:73-98)
  const size_t nf = tris.size() / 9;
  g.face_index.resize(nf * 3);
  for (size_t i = 0; i < nf * 3; i++) g.face_index[i] = (uint32_t)i;
  g.object_material = m;
  scene.geometry.push_back(std::move(g));
  if (m.lightsource) scene.light_sources.push_back(scene.geometry.size() - 1);
}

void add_floor(esc_scene &scene) {
  const std::vector<float> t = {-12, 0, 4,  12, 0, 4,   12,  0, -24,
                                -12, 0, 4,  12, 0, -24, -12, 0, -24};
  add_triangle_geometry(scene, t, matte(0.725f, 0.71f, 0.68f), "floor");
}

void add_light(esc_scene &scene) {
  Material m = matte(0.78f, 0.78f, 0.78f);
  m.ke[0] = 17.f;
  m.ke[1] = 12.f;
  m.ke[2] = 4.f;
  m.lightsource = true;
  // above the floor centroid (0,0,-10); winding makes the normal point down
  const std::vector<float> t = {-0.5f, 12, -9.5f, 0.f, 12, -10.5f, 0.5f, 12, -9.5f};
  add_triangle_geometry(scene, t, m, "light");
}

void add_spheres(esc_scene &scene, uint64_t seed, int n, float rlo, float rhi) {
  SplitMix64 rng{seed};
  for (int i = 0; i < n; i++) {
    Sphere s;
    s.cx = rng.uniform(-8.f, 8.f);
    s.cy = rng.uniform(0.5f, 5.f);
    s.cz = rng.uniform(-20.f, -2.f);
    s.r = rng.uniform(rlo, rhi);
    const float kr = rng.uniform(0.2f, 0.9f);
    const float kg = rng.uniform(0.2f, 0.9f);
    const float kb = rng.uniform(0.2f, 0.9f);
    scene.spheres.push_back(s);
    scene.sphere_materials.push_back(matte(kr, kg, kb));
  }
}

void add_heightfield(esc_scene &scene, int quads) {
  const float x0 = -12.f, x1 = 12.f, z0 = 4.f, z1 = -24.f;
  auto vx = [&](int i) { return x0 + (x1 - x0) * ((float)i / (float)quads); };
  auto vz = [&](int j) { return z0 + (z1 - z0) * ((float)j / (float)quads); };
  auto vy = [&](float x, float z) { return 0.4f * std::sin(0.9f * x) * std::cos(0.7f * z); };
  std::vector<float> t;
  t.reserve((size_t)quads * quads * 18);
  auto push = [&](int i, int j) {
    const float x = vx(i), z = vz(j);
    t.push_back(x);
    t.push_back(vy(x, z));
    t.push_back(z);
  };
  for (int j = 0; j < quads; j++) {
    for (int i = 0; i < quads; i++) {
      push(i, j); push(i + 1, j); push(i + 1, j + 1);
      push(i, j); push(i + 1, j + 1); push(i, j + 1);
    }
  }
  add_triangle_geometry(scene, t, matte(0.725f, 0.71f, 0.68f), "heightfield");
}

} // namespace

int make_synthetic(esc_scene &scene, const std::string &config, int n_override) {
  if (config == "c2" || config == "c3" || config == "c4") {
    add_floor(scene);
    add_light(scene);
    if (config == "c2") add_spheres(scene, 0xC2, n_override > 0 ? n_override : 100, 0.2f, 0.8f);
    if (config == "c3") add_spheres(scene, 0xC3, n_override > 0 ? n_override : 1000, 0.1f, 0.4f);
    if (config == "c4") add_spheres(scene, 0xC4, n_override > 0 ? n_override : 10000, 0.05f, 0.2f);
    return ESC_OK;
  }
  if (config == "c5") {
    add_heightfield(scene, n_override > 0 ? n_override : 224);
    add_light(scene);
    return ESC_OK;
  }
  set_error("esc_scene_synthetic: unknown config '" + config + "' (c2|c3|c4|c5)");
  return ESC_ERR_INVALID;
}

} // namespace esc
