// Host-side pieces of the C ABI (scene model, OBJ/MTL loader, synthetic scenes, flatten, PPM)
// exercised under AddressSanitizer + UBSan on the CPU.  Built and run by
// tests/test_host_surface.py::test_host_code_under_sanitizers; no HIP involved.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "accel_build.h"
#include "esctp1_rt.h"

// acceleration-structure builder (host/accel_build.cpp) on awkward inputs: every primitive must
// come out in exactly one leaf slot and the depth must stay inside the walk's stack
static int check_bvh(const std::vector<esc::PrimBox> &boxes, int block) {
  esc::BuiltBvh b;
  esc::build_bvh(boxes, block, 7u, esc::kBvhMaxDepth, b);
  std::vector<int> seen(boxes.size(), 0);
  for (int32_t k : b.order)
    if (k >= 0) {
      if ((size_t)k >= boxes.size()) return 1;
      seen[(size_t)k]++;
    }
  for (int c : seen)
    if (c != 1) return 1;
  if (b.depth > esc::kBvhMaxDepth) return 1;
  if (b.order.size() != (size_t)b.n_blocks * (size_t)block) return 1;
  for (const esc::BvhNode &n : b.nodes)
    for (int c = 0; c < 2; c++) {
      const int32_t ch = n.child[c];
      if (ch >= 0 ? (size_t)ch >= b.nodes.size() : (~ch) >= b.n_blocks) return 1;
    }
  return 0;
}

int main(int argc, char **argv) {
  int failures = 0;
  for (int i = 1; i < argc; i++) {
    esc_scene *sc = esc_scene_new();
    int rc = esc_scene_load_obj(sc, argv[i]);
    esc_scene_info info{};
    esc_scene_get_info(sc, &info);
    std::printf("%s rc=%d geoms=%d tris=%d lights=%d\n", argv[i], rc, info.n_geometry,
                info.n_triangles, info.n_lights);
    if (rc == ESC_OK) {
      for (int g = 0; g < info.n_geometry; g++) {
        int32_t cnt[3];
        esc_scene_geometry_counts(sc, g, cnt);
        std::vector<float> v((size_t)cnt[0] * 3), n((size_t)cnt[1] * 3 + 1);
        std::vector<uint32_t> f((size_t)cnt[2] * 3);
        float m[13];
        esc_scene_geometry_copy(sc, g, v.data(), n.data(), f.data(), m);
      }
      esc_flat_scene *flat = nullptr;
      if (esc_flatten_ispc(sc, 1, &flat) == ESC_OK) {
        int32_t nt = 0, nl = 0;
        esc_flat_triangles(flat, &nt);
        ispc_light *L = esc_flat_lights(flat, &nl);
        for (int l = 0; l < nl; l++)
          for (int k = 0; k < L[l].num_light_faces; k++) (void)L[l].light_faces[k];
        esc_flat_free(flat);
      }
    }
    esc_scene_free(sc);
  }
  for (const char *cfg : {"c2", "c3", "c5"}) {
    esc_scene *sc = esc_scene_new();
    if (esc_scene_synthetic(sc, cfg, std::strcmp(cfg, "c5") ? 0 : 16) != ESC_OK) failures++;
    esc_scene_free(sc);
  }
  {
    unsigned long long z = 12345;
    auto rnd = [&]() {
      z = z * 6364136223846793005ull + 1442695040888963407ull;
      return (float)((z >> 40) & 0xFFFF) / 65535.0f;
    };
    for (int n : {0, 1, 2, 3, 5, 64, 1000, 20000}) { // 20000: the threaded build
      std::vector<esc::PrimBox> scattered((size_t)n), same((size_t)n), line((size_t)n);
      float x = 1e-3f;
      for (int i = 0; i < n; i++) {
        for (int a = 0; a < 3; a++) {
          const float c = rnd() * 100.f - 50.f, h = rnd();
          scattered[(size_t)i].lo[a] = c - h;
          scattered[(size_t)i].hi[a] = c + h;
          same[(size_t)i].lo[a] = -1.f;
          same[(size_t)i].hi[a] = 1.f;
          line[(size_t)i].lo[a] = a ? 0.f : x;
          line[(size_t)i].hi[a] = a ? 0.f : x * 1.0001f;
        }
        x *= 1.05f; // geometric spacing: plain SAH would peel one box per level
      }
      for (int block : {2, 4})
        failures += check_bvh(scattered, block) + check_bvh(same, block) + check_bvh(line, block);
    }
  }
  esc_scene *bad = esc_scene_new();
  if (esc_scene_synthetic(bad, "nope", 0) == ESC_OK) failures++;
  if (esc_scene_load_obj(bad, "/nonexistent/file.obj") == ESC_OK) failures++;
  esc_scene_free(bad);
  // camera + PPM
  esc_camera cam;
  const float e[3] = {0, 1, 3}, l[3] = {0, 1, 0}, up[3] = {0, 1, 0};
  esc_camera_init(&cam, e, l, up, 60.f, 4.f / 3.f);
  std::vector<float> img(33 * 17 * 3);
  for (size_t i = 0; i < img.size(); i++) img[i] = (float)i / (float)img.size() * 1.3f;
  if (esc_write_ppm("/tmp/esc_sanitize.ppm", img.data(), 33, 17) != ESC_OK) failures++;
  std::printf("failures=%d\n", failures);
  return failures;
}
