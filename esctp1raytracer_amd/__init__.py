"""esctp1raytracer_amd -- MI355X-native drop-in for the per-pixel render loop of
pg42819/EscTp1RayTracer.

Python here is a thin host mirror over the C ABI (include/esctp1_rt.h, libesctp1rt.so);
all rendering happens in hand-written HIP kernels for gfx950.  Names follow the reference:
`Scene` ~ tracer::scene (scene.h), `Scene.load_obj` ~ model::loadobj (sceneloader.h:10),
`Camera` ~ tracer::camera (camera.h), `Renderer.render_rows` ~ the scan_row row loop
(main.cpp:628-636), `trace` ~ ispc::trace (trace.ispc:86-92), `write_ppm` ~ main.cpp:658-689.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (ESC_FACE_FIXED, ESC_FACE_HASH, ESC_STAGE_AUTO, ESC_STAGE_BVH, ESC_STAGE_LDS,
                    ESC_RENDER_EXACT_ONLY, ESC_RENDER_INDEX_ORDER, ESC_RENDER_SHADE_FUSED,
                    ESC_RENDER_SHADE_QUEUE, ESC_RENDER_TIME_KERNELS, ESC_RENDER_NO_TILE_LISTS, ESC_RENDER_NO_LIGHT_LISTS, ESC_RENDER_TWO_KERNELS, ESC_RENDER_BVH_HEURISTIC_PADS, ESC_RENDER_NO_COUNTERS,
                    ESC_STAGE_SMEM, EscError,
                    check)

__all__ = ["Scene", "Camera", "Renderer", "RecordedFrame", "FlatScene", "MultiRenderer", "render_multi", "render_multi_rccl", "rccl_available", "strip_local_rows", "trace", "write_ppm", "quantise", "synthetic_view",
           "EscError", "ESC_FACE_FIXED", "ESC_FACE_HASH", "ESC_STAGE_AUTO", "ESC_STAGE_SMEM",
           "ESC_STAGE_LDS", "ESC_STAGE_BVH", "ESC_RENDER_EXACT_ONLY", "ESC_RENDER_TIME_KERNELS", "ESC_RENDER_INDEX_ORDER", "ESC_RENDER_SHADE_QUEUE",
           "ESC_RENDER_SHADE_FUSED", "ESC_RENDER_NO_TILE_LISTS", "ESC_RENDER_NO_LIGHT_LISTS", "ESC_RENDER_TWO_KERNELS", "ESC_RENDER_BVH_HEURISTIC_PADS", "ESC_RENDER_NO_COUNTERS", "version"]


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def version():
    return _capi.load().esc_version().decode()


class Scene:
    """Host scene: geometries (de-indexed triangles + one material), light list, spheres."""

    def __init__(self):
        self._lib = _capi.load()
        self._h = self._lib.esc_scene_new()
        if not self._h:
            raise MemoryError("esc_scene_new")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.esc_scene_free(h)

    # -- building -------------------------------------------------------------------
    def add_geometry(self, vertex, face_index, material, normals=None):
        v = _f32(vertex, (-1, 3))
        f = np.ascontiguousarray(face_index, dtype=np.uint32).reshape(-1, 3)
        m = _f32(material, (13,))
        if normals is None or len(normals) == 0:
            n, nn = None, 0
        else:
            n = _f32(normals, (-1, 3))
            nn = n.shape[0]
        return check(self._lib.esc_scene_add_geometry(
            self._h, _fp(v), v.shape[0], _fp(n) if n is not None else None, nn,
            f.ctypes.data_as(C.POINTER(C.c_uint32)), f.shape[0], _fp(m)))

    def add_spheres(self, spheres_xyzr, materials):
        s = _f32(spheres_xyzr, (-1, 4))
        m = _f32(materials, (-1, 13))
        if s.shape[0] != m.shape[0]:
            raise ValueError("one material per sphere")
        check(self._lib.esc_scene_add_spheres(self._h, _fp(s), _fp(m), s.shape[0]))

    @classmethod
    def load_obj(cls, path):
        """model::loadobj (sceneloader.cpp:14-106)."""
        sc = cls()
        check(sc._lib.esc_scene_load_obj(sc._h, str(path).encode()))
        return sc

    @classmethod
    def synthetic(cls, config, n_override=0):
        """BASELINE.json workloads: 'c2' | 'c3' | 'c4' | 'c5' (SURVEY.md 8(d))."""
        sc = cls()
        check(sc._lib.esc_scene_synthetic(sc._h, config.encode(), int(n_override)))
        return sc

    # -- introspection --------------------------------------------------------------
    def info(self):
        i = _capi.esc_scene_info()
        check(self._lib.esc_scene_get_info(self._h, C.byref(i)))
        return {"n_geometry": i.n_geometry, "n_lights": i.n_lights,
                "n_triangles": i.n_triangles, "n_spheres": i.n_spheres}

    def geometry(self, g):
        cnt = (C.c_int32 * 3)()
        check(self._lib.esc_scene_geometry_counts(self._h, g, cnt))
        nv, nn, nf = cnt[0], cnt[1], cnt[2]
        v = np.zeros((nv, 3), np.float32)
        n = np.zeros((nn, 3), np.float32)
        f = np.zeros((nf, 3), np.uint32)
        m = np.zeros(13, np.float32)
        check(self._lib.esc_scene_geometry_copy(
            self._h, g, _fp(v), _fp(n) if nn else None,
            f.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(m)))
        return {"vertex": v, "normals": n, "face_index": f, "material": m}

    def light_sources(self):
        n = self.info()["n_lights"]
        out = np.zeros(max(n, 1), np.int32)
        check(self._lib.esc_scene_light_sources(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:n].copy()

    def spheres(self):
        n = self.info()["n_spheres"]
        s = np.zeros((n, 4), np.float32)
        m = np.zeros((n, 13), np.float32)
        if n:
            check(self._lib.esc_scene_spheres_copy(self._h, _fp(s), _fp(m)))
        return s, m

    def build_accel(self, origin, which):
        """Host-side build of the ESC_STAGE_BVH tree (no GPU needed): which = 'triangles' |
        'spheres'.  Returns nodes (structured array), order (block slot -> primitive index, -1 =
        pad), the padded primitive boxes (n, 2, 3) and the build summary."""
        w = {"triangles": 0, "spheres": 1}[which]
        o = _f32(origin, (3,))
        info = _capi.esc_accel_info()
        check(self._lib.esc_scene_build_accel(self._h, _fp(o), w, C.byref(info), None, 0, None, 0,
                                              None, 0))
        n_nodes = info.sph_nodes if w else info.tri_nodes
        n_blocks = info.sph_blocks if w else info.tri_blocks
        block = 4 if w else 2
        n_prims = self.info()["n_spheres" if w else "n_triangles"]
        node_dt = np.dtype([("lo0", np.float32, 3), ("hi0", np.float32, 3), ("lo1", np.float32, 3),
                            ("hi1", np.float32, 3), ("child", np.int32, 2),
                            ("minkey", np.uint32, 2)])
        nodes = np.zeros(max(n_nodes, 1), node_dt)
        order = np.full(max(n_blocks * block, 1), -1, np.int32)
        boxes = np.zeros((max(n_prims, 1), 2, 3), np.float32)
        check(self._lib.esc_scene_build_accel(
            self._h, _fp(o), w, C.byref(info),
            nodes.ctypes.data_as(C.POINTER(_capi.esc_bvh_node)), nodes.shape[0],
            order.ctypes.data_as(C.POINTER(C.c_int32)), order.shape[0], _fp(boxes),
            boxes.size))
        return {"nodes": nodes[:n_nodes], "order": order[:n_blocks * block],
                "boxes": boxes[:n_prims], "block": block,
                "root": info.sph_root if w else info.tri_root,
                "depth": info.sph_depth if w else info.tri_depth, "build_ms": info.build_ms}

    def flatten_ispc(self, sort_by_centroid_x=False):
        """flatten_scene_ispc (flatten_iscp.cpp:35-111) -> FlatScene."""
        return FlatScene(self, sort_by_centroid_x)


class FlatScene:
    """FlatScene of flatten_iscp.h:9-13: ispc_triangle[] / ispc_light[] / light faces."""

    def __init__(self, scene, sort_by_centroid_x=False):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        check(self._lib.esc_flatten_ispc(scene._h, 1 if sort_by_centroid_x else 0,
                                         C.byref(self._h)))
        n = C.c_int32()
        self.triangles = self._lib.esc_flat_triangles(self._h, C.byref(n))
        self.num_triangles = n.value
        self.light_triangles = self._lib.esc_flat_light_triangles(self._h, C.byref(n))
        self.num_light_triangles = n.value
        self.lights = self._lib.esc_flat_lights(self._h, C.byref(n))
        self.num_lights = n.value

    def check(self):
        """esc_check_flat: host-only validation of what `trace` would stage (raises EscError)."""
        check(self._lib.esc_check_flat(self.num_triangles, self.triangles, self.num_lights,
                                       self.lights, self.num_light_triangles,
                                       self.light_triangles))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.esc_flat_free(h)


class Camera:
    """tracer::camera (camera.h:16-29): the four vectors get_ray needs, computed on the host."""

    def __init__(self, lookfrom, lookat, vup=(0.0, 1.0, 0.0), vfov=60.0, aspect=4.0 / 3.0):
        self.lookfrom = _f32(lookfrom, (3,))
        self.lookat = _f32(lookat, (3,))
        self.vup = _f32(vup, (3,))
        self.vfov = float(np.float32(vfov))
        self.aspect = float(np.float32(aspect))
        self.c = _capi.esc_camera()
        _capi.load().esc_camera_init(C.byref(self.c), _fp(self.lookfrom), _fp(self.lookat),
                                     _fp(self.vup), self.vfov, self.aspect)

    @classmethod
    def for_image(cls, lookfrom, lookat, W, H, vup=(0.0, 1.0, 0.0), vfov=60.0):
        # main.cpp:548  float aspect = float(image_width) / image_height;
        return cls(lookfrom, lookat, vup, vfov, np.float32(W) / np.float32(H))

    def vectors(self):
        return {k: np.array(getattr(self.c, k), np.float32)
                for k in ("origin", "lower_left_corner", "horizontal", "vertical")}


def synthetic_view():
    eye = np.zeros(3, np.float32)
    look = np.zeros(3, np.float32)
    _capi.load().esc_synthetic_view(_fp(eye), _fp(look))
    return eye, look


def _options(shadows, face_mode, fixed_face, seed, stage, px=0, flags=0):
    o = _capi.esc_render_options()
    o.pixels_per_lane = px
    o.flags = flags
    o.shadows = 1 if shadows else 0
    o.face_mode = face_mode
    o.fixed_face = fixed_face
    o.stage = stage
    o.seed = seed
    return o


class Renderer:
    """Device context: one per GPU / host thread.  Raises EscError(ESC_ERR_NO_DEVICE) when no
    GPU is present -- there is no CPU fallback."""

    def __init__(self, device=0, stream=None):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        check(self._lib.esc_context_create(int(device), C.byref(self._h)))
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def __del__(self):
        self.close()

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.esc_context_destroy(h)

    def set_stream(self, stream):
        """stream: a raw hipStream_t (int) or an object with .cuda_stream (torch.cuda.Stream)."""
        ptr = getattr(stream, "cuda_stream", stream)
        check(self._lib.esc_context_set_stream(self._h, C.c_void_p(int(ptr))))

    def stream_handle(self):
        return self._lib.esc_context_stream(self._h)

    def synchronize(self):
        check(self._lib.esc_context_synchronize(self._h))

    def upload(self, scene):
        if isinstance(scene, FlatScene):
            check(self._lib.esc_upload_flat(self._h, scene.num_triangles, scene.triangles,
                                            scene.num_lights, scene.lights,
                                            scene.num_light_triangles, scene.light_triangles))
        else:
            check(self._lib.esc_upload_scene(self._h, scene._h))

    def render_rows(self, camera, W, H, row_begin, row_end, out_f32=None, out_u8=None, *,
                    shadows=True, face_mode=ESC_FACE_FIXED, fixed_face=0, seed=0,
                    stage=ESC_STAGE_AUTO, px=0, flags=0):
        """Asynchronous band render into DEVICE buffers (torch tensors or raw pointers),
        band-local layout ((h - row_begin) * W + w) * 3."""
        n = (row_end - row_begin) * W * 3

        def ptr(buf, itemsize):
            if buf is None:
                return None
            if hasattr(buf, "data_ptr"):
                if buf.numel() * buf.element_size() < n * itemsize:
                    raise ValueError("output buffer too small for the band")
                if not buf.is_cuda or not buf.is_contiguous():
                    raise ValueError("output must be a contiguous device tensor")
                return C.c_void_p(buf.data_ptr())
            return C.c_void_p(int(buf))

        o = _options(shadows, face_mode, fixed_face, seed, stage, px, flags)
        check(self._lib.esc_render_rows(self._h, C.byref(camera.c), W, H, row_begin, row_end,
                                        C.byref(o), ptr(out_f32, 4), ptr(out_u8, 1)))

    @staticmethod
    def _dev_ptr(buf, nbytes):
        if buf is None:
            return None
        if hasattr(buf, "data_ptr"):
            if buf.numel() * buf.element_size() < nbytes:
                raise ValueError("device buffer too small")
            if not buf.is_cuda or not buf.is_contiguous():
                raise ValueError("need a contiguous device tensor")
            return C.c_void_p(buf.data_ptr())
        return C.c_void_p(int(buf))

    def render_strips(self, camera, W, H, first_strip, strip_stride, out_f32=None, out_u8=None, *,
                      strip_rows=8, shadows=True, face_mode=ESC_FACE_FIXED, fixed_face=0, seed=0,
                      stage=ESC_STAGE_AUTO, px=0, flags=0):
        """Rank `first_strip` of `strip_stride`: renders strips first, first+stride, ... of
        `strip_rows` rows (counted from h = 0) into device buffers, rows packed in ascending h.
        Returns the number of rows rendered."""
        rows = strip_local_rows(H, strip_rows, first_strip, strip_stride)
        n = rows * W * 3
        o = _options(shadows, face_mode, fixed_face, seed, stage, px, flags)
        check(self._lib.esc_render_strips(self._h, C.byref(camera.c), W, H, strip_rows,
                                          first_strip, strip_stride, C.byref(o),
                                          self._dev_ptr(out_f32, n * 4), self._dev_ptr(out_u8, n)))
        return rows

    def record_strips(self, camera, W, H, first_strip, strip_stride, out_f32=None, out_u8=None, *,
                      strip_rows=8, shadows=True, face_mode=ESC_FACE_FIXED, fixed_face=0, seed=0,
                      stage=ESC_STAGE_AUTO, px=0, flags=0):
        """esc_frame_record: the launches of this render_strips call captured into a HIP graph.
        Returns a RecordedFrame; .launch() replays it with one host call."""
        rows = strip_local_rows(H, strip_rows, first_strip, strip_stride)
        n = rows * W * 3
        o = _options(shadows, face_mode, fixed_face, seed, stage, px, flags)
        h = C.c_void_p()
        check(self._lib.esc_frame_record(self._h, C.byref(camera.c), W, H, strip_rows, first_strip,
                                         strip_stride, C.byref(o), self._dev_ptr(out_f32, n * 4),
                                         self._dev_ptr(out_u8, n), C.byref(h)))
        return RecordedFrame(self._lib, h, (out_f32, out_u8))

    def assemble_strips(self, gathered, n_ranks, rank_pitch_bytes, W, H, frame, *, strip_rows=8,
                        bytes_per_pixel=12):
        """gathered: n_ranks blocks of local rows (block r at r*rank_pitch_bytes) -> frame."""
        check(self._lib.esc_assemble_strips(
            self._h, self._dev_ptr(gathered, 0), n_ranks, rank_pitch_bytes, W, H, strip_rows,
            bytes_per_pixel, self._dev_ptr(frame, W * H * bytes_per_pixel)))

    def render(self, camera, W, H, *, want_u8=False, shadows=True, face_mode=ESC_FACE_FIXED,
               fixed_face=0, seed=0, stage=ESC_STAGE_AUTO, px=0, flags=0, out=None):
        """Whole frame into host numpy arrays (synchronous): fp32 (H, W, 3), h = 0 bottom row,
        and optionally the PPM-quantised bytes.  `out`: a C-contiguous float32 (H, W, 3) array to
        render into (a fresh np.zeros of a 4K frame costs more in first-touch page faults than the
        frame and its copy back together)."""
        if out is not None:
            if out.dtype != np.float32 or out.shape != (H, W, 3) or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous float32 array of shape (H, W, 3)")
            img = out
        else:
            img = np.zeros((H, W, 3), np.float32)
        u8 = np.zeros((H, W, 3), np.uint8) if want_u8 else None
        o = _options(shadows, face_mode, fixed_face, seed, stage, px, flags)
        check(self._lib.esc_render_frame_host(
            self._h, C.byref(camera.c), W, H, C.byref(o), _fp(img),
            u8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_u8 else None))
        return (img, u8) if want_u8 else img

    def build_accel(self, origin):
        """Build (or rebuild) the ESC_STAGE_BVH tree now instead of at the first frame that
        asks for it."""
        o = _f32(origin, (3,))
        check(self._lib.esc_build_accel(self._h, _fp(o)))
        return self.accel_info()

    def accel_info(self):
        i = _capi.esc_accel_info()
        check(self._lib.esc_get_accel_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k in ("tri_nodes", "tri_blocks", "tri_depth", "tri_root",
                                           "sph_nodes", "sph_blocks", "sph_depth", "sph_root",
                                           "build_ms", "builds")}

    def last_kernel_ms(self):
        """(k_primary ms, k_shade ms) of the last frame rendered with ESC_RENDER_TIME_KERNELS."""
        ms = np.zeros(2, np.float32)
        check(self._lib.esc_last_kernel_ms(self._h, _fp(ms)))
        return float(ms[0]), float(ms[1])

    def reset_counters(self):
        check(self._lib.esc_reset_counters(self._h))

    def counters(self):
        c = _capi.esc_counters()
        check(self._lib.esc_read_counters(self._h, C.byref(c)))
        return {"primary_rays": c.primary_rays, "hit_pixels": c.hit_pixels,
                "shadow_rays": c.shadow_rays, "anyhit_tests": c.anyhit_tests,
                "anyhit_lane_tests": c.anyhit_lane_tests}


    def tile_lists(self, which):
        """the lists of the last frame (0 / 1: tile lists of spheres / triangles; 2 / 3: light lists of
        sphere / triangle pair records) -> dict with the
        header numbers and the per-tile counts, or None when the frame used none"""
        hdr = (C.c_int32 * 8)()
        n = check(self._lib.esc_tile_list_counts(self._h, which, hdr, None, 0))
        if n == 0:
            return None
        cnt = np.zeros(n, np.int32)
        check(self._lib.esc_tile_list_counts(self._h, which, hdr, cnt.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return {"global": hdr[0], "cones": hdr[1], "off": hdr[2], "tiles_x": hdr[3], "tile_rows": hdr[4],
                "cap": hdr[5], "global_cap": hdr[6], "counts": cnt.reshape(hdr[4], hdr[3])}


class RecordedFrame:
    """esc_frame: one frame's launches as a HIP graph on its renderer's stream"""

    def __init__(self, lib, handle, keep):
        self._lib, self._h, self._keep = lib, handle, keep  # the output buffers must outlive the graph
        self._launch = lib.esc_frame_launch

    def launch(self):
        rc = self._launch(self._h)
        if rc:
            check(rc)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.esc_frame_destroy(h)

    def __del__(self):
        self.close()


def strip_local_rows(H, strip_rows, first_strip, strip_stride):
    return check(_capi.load().esc_strip_local_rows(H, strip_rows, first_strip, strip_stride))


def render_multi(scene, camera, W, H, n_devices, *, want_u8=False, shadows=True,
                 face_mode=ESC_FACE_FIXED, fixed_face=0, seed=0, stage=ESC_STAGE_AUTO):
    """Single-process row-band split over n_devices (bands share devices when there are
    fewer GPUs than bands)."""
    lib = _capi.load()
    img = np.zeros((H, W, 3), np.float32)
    u8 = np.zeros((H, W, 3), np.uint8) if want_u8 else None
    ms = np.zeros(n_devices, np.float32)
    o = _options(shadows, face_mode, fixed_face, seed, stage)
    check(lib.esc_render_frame_multi(scene._h, C.byref(camera.c), W, H, C.byref(o), n_devices,
                                     _fp(img),
                                     u8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_u8 else None,
                                     _fp(ms)))
    return img, u8, ms


def rccl_available():
    """True when librccl.so could be bound at run time (esc_rccl_available)."""
    return bool(_capi.load().esc_rccl_available())


class MultiRenderer:
    """esc_multi: one process, n devices, 8-row strips dealt round-robin, the framebuffer gathered
    to the first device over RCCL (use_rccl=True) or by peer copies, then assembled there."""

    def __init__(self, n_devices, device_ids=None, use_rccl=True):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        ids = None
        if device_ids is not None:
            ids = (C.c_int32 * n_devices)(*[int(d) for d in device_ids])
        check(self._lib.esc_multi_create(int(n_devices), ids, 1 if use_rccl else 0,
                                         C.byref(self._h)))
        self.n_devices = int(n_devices)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.esc_multi_destroy(h)

    def __del__(self):
        self.close()

    def upload(self, scene):
        check(self._lib.esc_multi_upload_scene(self._h, scene._h))

    def render(self, camera, W, H, *, gather_u8=False, shadows=True, face_mode=ESC_FACE_FIXED,
               fixed_face=0, seed=0, stage=ESC_STAGE_AUTO, flags=0):
        """-> ((H, W, 3) fp32 or uint8 frame, per-device render ms, device address of the frame)"""
        o = _options(shadows, face_mode, fixed_face, seed, stage, 0, flags)
        ms = np.zeros(self.n_devices, np.float32)
        dptr = C.c_void_p()
        if gather_u8:
            out = np.zeros((H, W, 3), np.uint8)
            check(self._lib.esc_multi_render(self._h, C.byref(camera.c), W, H, C.byref(o), 1, None,
                                             out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                             C.byref(dptr), _fp(ms)))
        else:
            out = np.zeros((H, W, 3), np.float32)
            check(self._lib.esc_multi_render(self._h, C.byref(camera.c), W, H, C.byref(o), 0,
                                             _fp(out), None, C.byref(dptr), _fp(ms)))
        return out, ms, dptr.value


def render_multi_rccl(scene, camera, W, H, n_devices, *, want_u8=False, shadows=True,
                      face_mode=ESC_FACE_FIXED, fixed_face=0, seed=0, stage=ESC_STAGE_AUTO):
    """esc_render_frame_multi_rccl: the one-call form (communicator set up and torn down inside)."""
    lib = _capi.load()
    img = np.zeros((H, W, 3), np.float32)
    u8 = np.zeros((H, W, 3), np.uint8) if want_u8 else None
    ms = np.zeros(n_devices, np.float32)
    o = _options(shadows, face_mode, fixed_face, seed, stage)
    check(lib.esc_render_frame_multi_rccl(
        scene._h, C.byref(camera.c), W, H, C.byref(o), n_devices, _fp(img),
        u8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_u8 else None, _fp(ms)))
    return img, u8, ms


def trace(W, H, lookfrom, lookat, vup, vfov, aspect, flat, debug=0, test=0):
    """The ISPC drop-in symbol itself (trace.ispc:86-92 / main.cpp:619-624)."""
    lib = _capi.load()
    cam = _capi.ispc_cam()
    lib.esc_new_ispc_cam(C.byref(cam), _fp(_f32(lookfrom, (3,))), _fp(_f32(lookat, (3,))),
                         _fp(_f32(vup, (3,))), float(vfov), float(aspect))
    img = np.zeros((H, W, 3), np.float32)
    lib.trace(W, H, C.byref(cam), flat.num_triangles, flat.triangles, flat.num_lights,
              flat.lights, flat.num_light_triangles, flat.light_triangles, _fp(img), debug, test)
    return img


def quantise(image):
    img = _f32(image)
    out = np.zeros(img.shape, np.uint8)
    _capi.load().esc_quantise(_fp(img), img.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def write_ppm(path, image):
    """main.cpp:658-689 P3 writer; image (H, W, 3) fp32 or uint8, h = 0 bottom row."""
    lib = _capi.load()
    H, W = image.shape[0], image.shape[1]
    if image.dtype == np.uint8:
        a = np.ascontiguousarray(image)
        check(lib.esc_write_ppm_u8(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), W, H))
    else:
        a = _f32(image)
        check(lib.esc_write_ppm(str(path).encode(), _fp(a), W, H))
