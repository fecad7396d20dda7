// Host-side pieces of the C ABI (scene model, OBJ/MTL loader, synthetic scenes, flatten, PPM)
// exercised under AddressSanitizer + UBSan on the CPU.  Built and run by
// tests/test_host_surface.py::test_host_code_under_sanitizers; no HIP involved.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "esctp1_rt.h"

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
