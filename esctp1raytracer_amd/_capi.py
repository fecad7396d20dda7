"""ctypes binding of include/esctp1_rt.h (the C ABI of libesctp1rt.so).

This is the only way Python reaches the renderer.  There is no Python or CPU rendering
path: if the HIP library is missing this module raises, and if there is no GPU the render
entry points return ESC_ERR_NO_DEVICE, which `check` turns into an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ESC_LIB_PATH: developer override to time alternative builds of the same library side by side
LIB_PATH = os.environ.get("ESC_LIB_PATH") or os.path.join(_HERE, "lib", "libesctp1rt.so")

ESC_OK = 0
ESC_ERR_INVALID = -1
ESC_ERR_HIP = -2
ESC_ERR_IO = -3
ESC_ERR_PARSE = -4
ESC_ERR_NOMEM = -5
ESC_ERR_NO_DEVICE = -6
ESC_ERR_RCCL = -7

ESC_FACE_FIXED = 0
ESC_FACE_HASH = 1
ESC_STAGE_AUTO = 0
ESC_STAGE_SMEM = 1
ESC_STAGE_LDS = 2
ESC_STAGE_BVH = 3
ESC_MATERIAL_FLOATS = 13
ESC_RENDER_EXACT_ONLY = 1
ESC_RENDER_TIME_KERNELS = 2
ESC_RENDER_INDEX_ORDER = 4
ESC_RENDER_SHADE_QUEUE = 8
ESC_RENDER_SHADE_FUSED = 16
ESC_RENDER_NO_TILE_LISTS = 32
ESC_RENDER_NO_LIGHT_LISTS = 64
ESC_RENDER_TWO_KERNELS = 128
ESC_RENDER_BVH_HEURISTIC_PADS = 256
ESC_RENDER_NO_COUNTERS = 512


class EscError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"esctp1_rt error {code}: {message}")
        self.code = code


class esc_scene_info(C.Structure):
    _fields_ = [("n_geometry", C.c_int32), ("n_lights", C.c_int32),
                ("n_triangles", C.c_int32), ("n_spheres", C.c_int32)]


class esc_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3)]


class ispc_triangle(C.Structure):  # ispc_helpers.h:16-29
    _fields_ = [("vertices", (C.c_float * 3) * 3), ("normals", (C.c_float * 3) * 3),
                ("prim_id", C.c_int32), ("geom_id", C.c_int32),
                ("has_normals", C.c_int32), ("is_light", C.c_int32),
                ("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3),
                ("ke", C.c_float * 3), ("Ns", C.c_float)]


class ispc_light(C.Structure):  # ispc_helpers.h:52-56
    _fields_ = [("geom_id", C.c_int32), ("light_faces", C.POINTER(C.c_int32)),
                ("num_light_faces", C.c_int32)]


class ispc_cam(C.Structure):  # ispc_helpers.h:59-65
    _fields_ = [("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3),
                ("vup", C.c_float * 3), ("vfov", C.c_float), ("aspect", C.c_float)]


class esc_render_options(C.Structure):
    _fields_ = [("shadows", C.c_int32), ("face_mode", C.c_int32), ("fixed_face", C.c_int32),
                ("stage", C.c_int32), ("seed", C.c_uint64), ("pixels_per_lane", C.c_int32),
                ("flags", C.c_int32)]


class esc_counters(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("hit_pixels", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("anyhit_tests", C.c_uint64),
                ("anyhit_lane_tests", C.c_uint64)]


class esc_bvh_node(C.Structure):  # 64 bytes
    _fields_ = [("lo0", C.c_float * 3), ("hi0", C.c_float * 3), ("lo1", C.c_float * 3),
                ("hi1", C.c_float * 3), ("child", C.c_int32 * 2), ("minkey", C.c_uint32 * 2)]


class esc_accel_info(C.Structure):
    _fields_ = [("tri_nodes", C.c_int32), ("tri_blocks", C.c_int32), ("tri_depth", C.c_int32),
                ("tri_root", C.c_int32), ("sph_nodes", C.c_int32), ("sph_blocks", C.c_int32),
                ("sph_depth", C.c_int32), ("sph_root", C.c_int32), ("build_ms", C.c_float),
                ("builds", C.c_int32), ("reserved", C.c_int32 * 2)]


_P = C.c_void_p
_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)
_U32 = C.POINTER(C.c_uint32)

# name -> (restype, argtypes); every symbol include/esctp1_rt.h declares
SIGNATURES = {
    "esc_last_error": (C.c_char_p, []),
    "esc_version": (C.c_char_p, []),
    "esc_scene_new": (_P, []),
    "esc_scene_free": (None, [_P]),
    "esc_scene_add_geometry": (C.c_int, [_P, _F, C.c_int32, _F, C.c_int32, _U32, C.c_int32, _F]),
    "esc_scene_add_spheres": (C.c_int, [_P, _F, _F, C.c_int32]),
    "esc_scene_load_obj": (C.c_int, [_P, C.c_char_p]),
    "esc_scene_synthetic": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "esc_synthetic_view": (None, [_F, _F]),
    "esc_scene_get_info": (C.c_int, [_P, C.POINTER(esc_scene_info)]),
    "esc_scene_geometry_counts": (C.c_int, [_P, C.c_int32, _I32]),
    "esc_scene_geometry_copy": (C.c_int, [_P, C.c_int32, _F, _F, _U32, _F]),
    "esc_scene_light_sources": (C.c_int, [_P, _I32]),
    "esc_scene_spheres_copy": (C.c_int, [_P, _F, _F]),
    "esc_camera_init": (None, [C.POINTER(esc_camera), _F, _F, _F, C.c_float, C.c_float]),
    "esc_flatten_ispc": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    "esc_flat_free": (None, [_P]),
    "esc_flat_triangles": (C.POINTER(ispc_triangle), [_P, _I32]),
    "esc_flat_light_triangles": (C.POINTER(ispc_triangle), [_P, _I32]),
    "esc_flat_lights": (C.POINTER(ispc_light), [_P, _I32]),
    "esc_new_ispc_cam": (None, [C.POINTER(ispc_cam), _F, _F, _F, C.c_float, C.c_float]),
    "trace": (None, [C.c_int32, C.c_int32, C.POINTER(ispc_cam), C.c_int32,
                     C.POINTER(ispc_triangle), C.c_int32, C.POINTER(ispc_light), C.c_int32,
                     C.POINTER(ispc_triangle), _F, C.c_int32, C.c_int32]),
    "esc_context_create": (C.c_int, [C.c_int32, C.POINTER(_P)]),
    "esc_context_destroy": (None, [_P]),
    "esc_context_set_stream": (C.c_int, [_P, _P]),
    "esc_context_stream": (_P, [_P]),
    "esc_context_synchronize": (C.c_int, [_P]),
    "esc_upload_scene": (C.c_int, [_P, _P]),
    "esc_upload_flat": (C.c_int, [_P, C.c_int32, C.POINTER(ispc_triangle), C.c_int32,
                                  C.POINTER(ispc_light), C.c_int32, C.POINTER(ispc_triangle)]),
    "esc_check_flat": (C.c_int, [C.c_int32, C.POINTER(ispc_triangle), C.c_int32,
                                 C.POINTER(ispc_light), C.c_int32, C.POINTER(ispc_triangle)]),
    "esc_render_rows": (C.c_int, [_P, C.POINTER(esc_camera), C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.POINTER(esc_render_options), _P, _P]),
    "esc_render_strips": (C.c_int, [_P, C.POINTER(esc_camera), C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.POINTER(esc_render_options), _P, _P]),
    "esc_frame_record": (C.c_int, [_P, C.POINTER(esc_camera), C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.POINTER(esc_render_options), _P, _P,
                                   C.POINTER(C.c_void_p)]),
    "esc_frame_launch": (C.c_int, [_P]),
    "esc_frame_destroy": (None, [_P]),
    "esc_strip_local_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "esc_assemble_strips": (C.c_int, [_P, _P, C.c_int32, C.c_size_t, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, _P]),
    "esc_build_accel": (C.c_int, [_P, _F]),
    "esc_get_accel_info": (C.c_int, [_P, C.POINTER(esc_accel_info)]),
    "esc_scene_build_accel": (C.c_int, [_P, _F, C.c_int32, C.POINTER(esc_accel_info),
                                        C.POINTER(esc_bvh_node), C.c_int64, _I32, C.c_int64, _F,
                                        C.c_int64]),
    "esc_queue_schedule": (C.c_int, [C.c_int32, C.c_int32, _I32, C.c_int32]),
    "esc_tri_group_record": (C.c_int, [_F, C.c_int32, _F]),
    "esc_sphere_group_record": (C.c_int, [_F, C.c_int32, _F]),
    "esc_tile_list_counts": (C.c_int, [C.c_void_p, C.c_int32, _I32, _I32, C.c_size_t]),
    "esc_tile_list_ids": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, _I32, C.c_int32]),
    "esc_tile_rect": (C.c_int, [C.POINTER(esc_camera), C.c_int32, C.c_int32, _F, C.c_double, _I32]),
    "esc_tile_band": (C.c_int, [C.POINTER(esc_camera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F,
                                C.c_double]),
    "esc_group_order": (C.c_int, [_F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _I32]),
    "esc_last_kernel_ms": (C.c_int, [_P, _F]),
    "esc_reset_counters": (C.c_int, [_P]),
    "esc_read_counters": (C.c_int, [_P, C.POINTER(esc_counters)]),
    "esc_render_frame_host": (C.c_int, [_P, C.POINTER(esc_camera), C.c_int32, C.c_int32,
                                        C.POINTER(esc_render_options), _F, _U8]),
    "esc_render_frame_multi": (C.c_int, [_P, C.POINTER(esc_camera), C.c_int32, C.c_int32,
                                         C.POINTER(esc_render_options), C.c_int32, _F, _U8, _F]),
    "esc_rccl_available": (C.c_int, []),
    "esc_multi_create": (C.c_int, [C.c_int32, _I32, C.c_int32, C.POINTER(_P)]),
    "esc_multi_destroy": (None, [_P]),
    "esc_multi_upload_scene": (C.c_int, [_P, _P]),
    "esc_multi_render": (C.c_int, [_P, C.POINTER(esc_camera), C.c_int32, C.c_int32,
                                   C.POINTER(esc_render_options), C.c_int32, _F, _U8,
                                   C.POINTER(_P), _F]),
    "esc_render_frame_multi_rccl": (C.c_int, [_P, C.POINTER(esc_camera), C.c_int32, C.c_int32,
                                              C.POINTER(esc_render_options), C.c_int32, _F, _U8,
                                              _F]),
    "esc_write_ppm": (C.c_int, [C.c_char_p, _F, C.c_int32, C.c_int32]),
    "esc_write_ppm_u8": (C.c_int, [C.c_char_p, _U8, C.c_int32, C.c_int32]),
    "esc_quantise": (None, [_F, C.c_int64, _U8]),
}

_lib = None


def load():
    """dlopen libesctp1rt.so and type every entry point.  Raises if the library is missing:
    the HIP extension is the product, there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 and preloads it by
    # path.  If ours (resolved to /opt/rocm) is loaded first the process ends up with two
    # runtimes and the second one sees no GPU.  Loading torch first makes our NEEDED entry bind
    # to the copy torch already mapped.  (The C++ viewer has no torch and uses /opt/rocm's.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make lib` (or __graft_entry__.build()); "
            "esctp1raytracer_amd has no non-HIP fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise EscError(rc, load().esc_last_error().decode("utf-8", "replace"))
    return rc
