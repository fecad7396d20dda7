// rt_multi.cpp -- single-process multi-GPU frames behind the C ABI:
// the caller of the reference's row loop (main.cpp:628-636 renders rows independently) hands over
// `n_devices`; every device renders its share of the 8-row strips and the ONE exchange step --
// the framebuffer gather to device 0 -- runs over RCCL (ncclSend / ncclRecv in one group: direct
// peer -> root transfers, one xGMI link per peer), then k_assemble_strips lays the frame out.
//
// RCCL is bound at run time (dlopen), not at link time: libesctp1rt.so stays loadable on a host
// without RCCL and, inside a PyTorch process, binds the librccl.so torch already mapped instead of
// bringing a second copy.  Only <rccl/rccl.h>'s types are used at compile time.
//
// This file only talks to the rest of the library through include/esctp1_rt.h (contexts, strips,
// assemble), so the partition arithmetic is the one the per-rank path (bench.py, multigpu.py)
// uses and the GPU tests cover.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../host/scene.h"

using esc::set_error;

namespace {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string why; // set when loading failed
};

const RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *env = std::getenv("ESC_RCCL_LIB");
    const char *names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    // a copy that is already mapped (torch's) first, then the system one
    for (int pass = 0; pass < 2 && !api.handle; pass++)
      for (const char *n : names) {
        if (!n || !*n) continue;
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (api.handle) break;
      }
    if (!api.handle) {
      const char *e = dlerror();
      api.why = std::string("RCCL not available: ") + (e ? e : "librccl.so not found");
      return;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(api.handle, n);
      if (!p && api.why.empty()) api.why = std::string("RCCL symbol missing: ") + n;
      return p;
    };
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  return api;
}

} // namespace

struct esc_multi {
  int n = 0;
  bool use_rccl = false;
  std::vector<int> devices;
  std::vector<esc_context *> ctx;
  std::vector<ncclComm_t> comm; // use_rccl only
  // per-device packed strips (device i renders into out[i]); device 0 renders straight into block 0
  // of `gathered`, which also receives the peers' blocks
  std::vector<void *> out;
  void *gathered = nullptr, *frame = nullptr;
  size_t out_cap = 0, gathered_cap = 0, frame_cap = 0;
  std::vector<hipEvent_t> t0, t1, t2; // render start / end, peer copy done
  bool have_scene = false;
};

namespace {

constexpr int kStrip = 8;

#define M_HIP(expr)                                                          \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
      return ESC_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)
#define M_RCCL(expr)                                                         \
  do {                                                                       \
    ncclResult_t r_ = (expr);                                                \
    if (r_ != ncclSuccess) {                                                 \
      set_error(std::string(#expr) + ": " + rccl().GetErrorString(r_));      \
      return ESC_ERR_RCCL;                                                   \
    }                                                                        \
  } while (0)

int max_local_rows(int H, int n) { return esc_strip_local_rows(H, kStrip, 0, n); } // rank 0 has most

int grow(void *&p, size_t &cap, size_t need) {
  if (cap >= need) return ESC_OK;
  if (p) M_HIP(hipFree(p));
  p = nullptr;
  cap = 0;
  M_HIP(hipMalloc(&p, need));
  cap = need;
  return ESC_OK;
}

} // namespace

extern "C" {

int esc_rccl_available(void) {
  const RcclApi &a = rccl();
  if (a.handle && a.why.empty()) return 1;
  set_error(a.why);
  return 0;
}

void esc_multi_destroy(esc_multi *m) {
  if (!m) return;
  for (int i = 0; i < (int)m->ctx.size(); i++) {
    if (!m->ctx[i]) continue;
    (void)hipSetDevice(m->devices[i]);
    (void)esc_context_synchronize(m->ctx[i]);
    if (i < (int)m->comm.size() && m->comm[i]) (void)rccl().CommDestroy(m->comm[i]);
    if (i < (int)m->out.size() && m->out[i] && i != 0) (void)hipFree(m->out[i]);
    if (i < (int)m->t0.size() && m->t0[i]) (void)hipEventDestroy(m->t0[i]);
    if (i < (int)m->t1.size() && m->t1[i]) (void)hipEventDestroy(m->t1[i]);
    if (i < (int)m->t2.size() && m->t2[i]) (void)hipEventDestroy(m->t2[i]);
    if (i == 0) {
      if (m->gathered) (void)hipFree(m->gathered);
      if (m->frame) (void)hipFree(m->frame);
    }
    esc_context_destroy(m->ctx[i]);
  }
  delete m;
}

int esc_multi_create(int32_t n_devices, const int32_t *device_ids, int32_t use_rccl,
                     esc_multi **out) {
  if (!out || n_devices < 1) {
    set_error("esc_multi_create: bad argument");
    return ESC_ERR_INVALID;
  }
  int avail = 0;
  if (hipGetDeviceCount(&avail) != hipSuccess || avail < 1) {
    set_error("esc_multi_create: no HIP device (this renderer has no CPU fallback)");
    return ESC_ERR_NO_DEVICE;
  }
  // With RCCL every rank needs its own device (one communicator rank per device).  WITHOUT it
  // (use_rccl == 0: peer copies) ranks may share devices -- device_ids may repeat, and with
  // device_ids == NULL rank i takes device i % count: the whole n-rank path (partition, per-rank
  // pitch and offsets, copies into the gathered blocks, assembly) then runs on however many GPUs
  // there are.  That is how the 1-GPU test box exercises n = 2, 3, 8.
  if (use_rccl && n_devices > avail) {
    set_error("esc_multi_create: n_devices exceeds the device count (one RCCL rank per device; "
              "use_rccl = 0 lets ranks share devices)");
    return ESC_ERR_INVALID;
  }
  esc_multi *m = new (std::nothrow) esc_multi();
  if (!m) return ESC_ERR_NOMEM;
  m->n = n_devices;
  m->use_rccl = use_rccl != 0;
  for (int i = 0; i < n_devices; i++) {
    const int d = device_ids ? device_ids[i] : (use_rccl ? i : i % avail);
    if (d < 0 || d >= avail || (use_rccl && std::count(m->devices.begin(), m->devices.end(), d))) {
      set_error("esc_multi_create: device ids must be in range, and distinct with RCCL");
      delete m;
      return ESC_ERR_INVALID;
    }
    m->devices.push_back(d);
  }
  m->ctx.assign((size_t)n_devices, nullptr);
  m->out.assign((size_t)n_devices, nullptr);
  m->t0.assign((size_t)n_devices, nullptr);
  m->t1.assign((size_t)n_devices, nullptr);
  m->t2.assign((size_t)n_devices, nullptr);
  for (int i = 0; i < n_devices; i++) {
    int rc = esc_context_create(m->devices[i], &m->ctx[i]);
    if (rc == ESC_OK) {
      if (hipEventCreate(&m->t0[i]) != hipSuccess || hipEventCreate(&m->t1[i]) != hipSuccess ||
          hipEventCreate(&m->t2[i]) != hipSuccess) {
        set_error("esc_multi_create: hipEventCreate failed");
        rc = ESC_ERR_HIP;
      }
    }
    if (rc != ESC_OK) {
      esc_multi_destroy(m);
      return rc;
    }
  }
  if (m->use_rccl) {
    if (!esc_rccl_available()) {
      esc_multi_destroy(m);
      return ESC_ERR_RCCL;
    }
    m->comm.assign((size_t)n_devices, nullptr);
    ncclResult_t r = rccl().CommInitAll(m->comm.data(), n_devices, m->devices.data());
    if (r != ncclSuccess) {
      set_error(std::string("ncclCommInitAll: ") + rccl().GetErrorString(r));
      m->comm.clear();
      esc_multi_destroy(m);
      return ESC_ERR_RCCL;
    }
  }
  *out = m;
  return ESC_OK;
}

int esc_multi_upload_scene(esc_multi *m, const esc_scene *scene) {
  if (!m || !scene) {
    set_error("esc_multi_upload_scene: bad argument");
    return ESC_ERR_INVALID;
  }
  for (int i = 0; i < m->n; i++) { // the scene is replicated: every device tests every primitive
    int rc = esc_upload_scene(m->ctx[i], scene);
    if (rc) return rc;
  }
  m->have_scene = true;
  return ESC_OK;
}

int esc_multi_render(esc_multi *m, const esc_camera *cam, int32_t W, int32_t H,
                     const esc_render_options *opts, int32_t gather_u8, float *image,
                     uint8_t *rgb8, void **d_frame, float *ms_per_device) {
  if (!m || !cam || !opts || (gather_u8 && image) || (!gather_u8 && rgb8)) {
    set_error("esc_multi_render: bad argument (gather_u8 selects which of image / rgb8 may be set)");
    return ESC_ERR_INVALID;
  }
  if (!m->have_scene) {
    set_error("esc_multi_render: no scene uploaded");
    return ESC_ERR_INVALID;
  }
  if (W < 2 || H < 2) {
    set_error("esc_multi_render: need W,H >= 2");
    return ESC_ERR_INVALID;
  }
  const int n = m->n;
  const size_t bpp = gather_u8 ? 3 : 12; // bytes per pixel of what is gathered
  const size_t pitch = (size_t)max_local_rows(H, n) * W * bpp; // every rank's block, padded alike
  const size_t frame_bytes = (size_t)W * H * bpp;
  int rc;
  M_HIP(hipSetDevice(m->devices[0]));
  M_HIP(hipStreamSynchronize((hipStream_t)esc_context_stream(m->ctx[0]))); // buffers may be regrown
  if ((rc = grow(m->gathered, m->gathered_cap, pitch * n))) return rc;
  if ((rc = grow(m->frame, m->frame_cap, frame_bytes))) return rc;
  m->out[0] = m->gathered;
  for (int i = 1; i < n; i++) {
    M_HIP(hipSetDevice(m->devices[i]));
    if (m->out_cap < pitch) {
      M_HIP(hipStreamSynchronize((hipStream_t)esc_context_stream(m->ctx[i])));
      if (m->out[i]) M_HIP(hipFree(m->out[i]));
      m->out[i] = nullptr;
      M_HIP(hipMalloc(&m->out[i], pitch));
    }
  }
  m->out_cap = std::max(m->out_cap, pitch);

  // every device renders its strips (rank i of n); nothing is waited on until all are launched
  std::vector<size_t> bytes((size_t)n);
  for (int i = 0; i < n; i++) {
    hipStream_t st = (hipStream_t)esc_context_stream(m->ctx[i]);
    M_HIP(hipSetDevice(m->devices[i]));
    M_HIP(hipEventRecord(m->t0[i], st));
    rc = esc_render_strips(m->ctx[i], cam, W, H, kStrip, i, n, opts,
                           gather_u8 ? nullptr : (float *)m->out[i],
                           gather_u8 ? (uint8_t *)m->out[i] : nullptr);
    if (rc) return rc;
    M_HIP(hipEventRecord(m->t1[i], st));
    bytes[(size_t)i] = (size_t)esc_strip_local_rows(H, kStrip, i, n) * W * bpp;
  }
  // the ONE exchange step: peers send their packed strips to device 0
  if (n > 1) {
    hipStream_t st0 = (hipStream_t)esc_context_stream(m->ctx[0]);
    if (m->use_rccl) {
      const RcclApi &R = rccl();
      M_RCCL(R.GroupStart());
      for (int i = 1; i < n; i++) {
        if (!bytes[(size_t)i]) continue;
        ncclResult_t r = R.Send(m->out[i], bytes[(size_t)i], ncclUint8, 0, m->comm[i],
                                (hipStream_t)esc_context_stream(m->ctx[i]));
        if (r == ncclSuccess)
          r = R.Recv((char *)m->gathered + (size_t)i * pitch, bytes[(size_t)i], ncclUint8, i,
                     m->comm[0], st0);
        if (r != ncclSuccess) {
          (void)R.GroupEnd();
          set_error(std::string("ncclSend/ncclRecv: ") + R.GetErrorString(r));
          return ESC_ERR_RCCL;
        }
      }
      M_RCCL(R.GroupEnd());
    } else { // peer copies ordered after each peer's render, then joined into device 0's stream
      for (int i = 1; i < n; i++) {
        if (!bytes[(size_t)i]) continue;
        hipStream_t sti = (hipStream_t)esc_context_stream(m->ctx[i]);
        M_HIP(hipSetDevice(m->devices[i]));
        if (m->devices[i] == m->devices[0]) // ranks sharing a device: a plain device-to-device copy
          M_HIP(hipMemcpyAsync((char *)m->gathered + (size_t)i * pitch, m->out[i], bytes[(size_t)i],
                               hipMemcpyDeviceToDevice, sti));
        else
          M_HIP(hipMemcpyPeerAsync((char *)m->gathered + (size_t)i * pitch, m->devices[0], m->out[i],
                                   m->devices[i], bytes[(size_t)i], sti));
        M_HIP(hipEventRecord(m->t2[i], sti)); // (t1 keeps bracketing the render: ms_per_device)
        M_HIP(hipSetDevice(m->devices[0]));
        M_HIP(hipStreamWaitEvent(st0, m->t2[i], 0));
      }
    }
  }
  M_HIP(hipSetDevice(m->devices[0]));
  hipStream_t st0 = (hipStream_t)esc_context_stream(m->ctx[0]);
  rc = esc_assemble_strips(m->ctx[0], m->gathered, n, pitch, W, H, kStrip, (int32_t)bpp, m->frame);
  if (rc) return rc;
  void *host = gather_u8 ? (void *)rgb8 : (void *)image;
  if (host) M_HIP(hipMemcpyAsync(host, m->frame, frame_bytes, hipMemcpyDeviceToHost, st0));
  for (int i = 0; i < n; i++) {
    M_HIP(hipSetDevice(m->devices[i]));
    M_HIP(hipStreamSynchronize((hipStream_t)esc_context_stream(m->ctx[i])));
    if (ms_per_device) {
      ms_per_device[i] = 0.f;
      M_HIP(hipEventElapsedTime(&ms_per_device[i], m->t0[i], m->t1[i]));
    }
  }
  if (d_frame) *d_frame = m->frame;
  return ESC_OK;
}

int esc_render_frame_multi_rccl(const esc_scene *scene, const esc_camera *cam, int32_t W, int32_t H,
                                const esc_render_options *opts, int32_t n_devices, float *image,
                                uint8_t *rgb8, float *ms_per_device) {
  if (!scene || !cam || !opts || n_devices < 1 || (!image && !rgb8)) {
    set_error("esc_render_frame_multi_rccl: bad argument");
    return ESC_ERR_INVALID;
  }
  esc_multi *m = nullptr;
  int rc = esc_multi_create(n_devices, nullptr, /*use_rccl=*/1, &m);
  if (rc) return rc;
  rc = esc_multi_upload_scene(m, scene);
  // fp32 is what the `trace` seam returns; the bytes are what the PPM writer needs (main.cpp:676-682)
  if (rc == ESC_OK && image) rc = esc_multi_render(m, cam, W, H, opts, 0, image, nullptr, nullptr, ms_per_device);
  if (rc == ESC_OK && rgb8) rc = esc_multi_render(m, cam, W, H, opts, 1, nullptr, rgb8, nullptr, image ? nullptr : ms_per_device);
  esc_multi_destroy(m);
  return rc;
}

} // extern "C"
