"""ctypes view of include/rt_amd.h (libraytracer_amd.so).

Loading fails loudly when the library has not been built: there is no CPU or
pure-Python fallback for any compute entry point.
"""
import ctypes as C
import os

from . import build as _build

vp = C.c_void_p


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rt_status {status}: {message}")
        self.status = status
        self.message = message


class Mesh(C.Structure):
    _fields_ = [("verts", vp), ("material", vp), ("group_index", vp), ("texcoords", vp),
                ("n_tris", C.c_int32), ("kd", vp), ("is_light", vp), ("texture", vp),
                ("n_materials", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("size", C.c_float * 2),
                ("emission", C.c_float * 3), ("normal", C.c_float * 3), ("found", C.c_int32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_tris", C.c_int32), ("n_nodes", C.c_int32), ("max_depth", C.c_int32),
                ("n_materials", C.c_int32), ("device", C.c_int32), ("has_light", C.c_int32),
                ("node_bytes", C.c_uint64), ("tri_bytes", C.c_uint64),
                ("total_device_bytes", C.c_uint64), ("ordered_bytes", C.c_uint64),
                ("linked_view", C.c_int32),
                ("ordered_view", C.c_int32),
                ("max_paths_in_flight", C.c_uint64), ("workspace_bytes", C.c_uint64),
                ("near_lo", C.c_float * 3), ("near_hi", C.c_float * 3)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("box_tests", C.c_uint64),
                ("tris_fetched", C.c_uint64), ("tris_tested", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("kernel_launches", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Camera(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("projection_plane_distance", C.c_float),
                ("rays_per_pixel", C.c_int32), ("projection", C.c_int32),
                ("field_of_view", C.c_float), ("ortographic_size", C.c_float)]


class Settings(C.Structure):
    _fields_ = [("draw_surface_normal", C.c_int32), ("gamma_correction", C.c_int32),
                ("gamma", C.c_float), ("light_sample_rays", C.c_int32),
                ("bounce_limit", C.c_int32), ("light_intensity", C.c_float),
                ("debug_light_size", C.c_float), ("force_debug_light", C.c_int32),
                ("debug_light_at_camera", C.c_int32), ("debug_light_transform", C.c_float * 16)]


class Pass(C.Structure):
    _fields_ = [("camera", Camera), ("total_width", C.c_int32), ("total_height", C.c_int32),
                ("task_id", C.c_int32), ("pad0", C.c_int32), ("rng_seed", C.c_int64),
                ("x_offset", C.c_int32), ("y_offset", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("settings", Settings), ("render_key", C.c_int32),
                ("pad1", C.c_int32)]


# every symbol include/rt_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_bvh_build": (C.c_int32, [C.POINTER(Mesh), C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "rt_bvh_from_json": (C.c_int32, [C.POINTER(Mesh), C.c_char_p, C.c_size_t, C.POINTER(vp)]),
    "rt_bvh_to_json": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "rt_bvh_save": (C.c_int32, [vp, C.c_char_p]),
    "rt_bvh_load": (C.c_int32, [C.POINTER(Mesh), C.c_char_p, C.POINTER(vp)]),
    "rt_bvh_node_count": (C.c_int32, [vp]),
    "rt_bvh_tri_count": (C.c_int32, [vp]),
    "rt_bvh_max_depth": (C.c_int32, [vp]),
    "rt_bvh_export": (C.c_int32, [vp] * 9),
    "rt_bvh_destroy": (None, [vp]),
    "rt_free": (None, [vp]),
    "rt_flatten": (C.c_int32, [C.POINTER(Mesh), vp, vp, vp, vp]),
    "rt_flatten_ordered": (C.c_int32, [C.POINTER(Mesh), vp, vp, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "rt_mesh_light": (C.c_int32, [C.POINTER(Mesh), vp, C.POINTER(Light)]),
    "rt_scene_create": (C.c_int32, [C.POINTER(Mesh), vp, C.c_int32, C.POINTER(vp)]),
    "rt_scene_info_get": (C.c_int32, [vp, C.POINTER(SceneInfo)]),
    "rt_scene_light": (C.c_int32, [vp, C.POINTER(Light)]),
    "rt_scene_set_sample_tables": (C.c_int32, [vp, vp, C.c_int32, vp, C.c_int32]),
    "rt_default_sample_table": (C.c_int32, [C.c_int32, C.c_uint64, vp]),
    "rt_png_decode": (C.c_int32, [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32)]),
    "rt_scene_set_texture": (C.c_int32, [vp, C.c_int32, vp, C.c_int32, C.c_int32]),
    "rt_scene_set_max_paths": (C.c_int32, [vp, C.c_uint64]),
    "rt_scene_set_workspace_limit": (C.c_int32, [vp, C.c_uint64]),
    "rt_workspace_bytes_per_path": (C.c_uint64, [C.c_int32, C.c_int32]),
    "rt_scene_set_tuning": (C.c_int32, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rt_scene_set_counters": (C.c_int32, [vp, C.c_int32]),
    "rt_scene_destroy": (None, [vp]),
    "rt_trace_rays": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_int32, C.c_int32, C.POINTER(Stats)]),
    "rt_trace_rays_device": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_int32, vp]),
    "rt_trace_rays_device_counted": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_int32, vp, vp]),
    "rt_camera_setup": (C.c_int32, [C.POINTER(Camera), C.c_int32, C.c_int32, vp, vp]),
    "rt_pass_params": (C.c_int32, [C.POINTER(Light), C.POINTER(Pass), vp, C.c_size_t]),
    "rt_render_tile": (C.c_int32, [vp, C.POINTER(Pass), C.c_int32, vp, vp, C.POINTER(Stats)]),
    "rt_render_tile_device": (C.c_int32, [vp, C.POINTER(Pass), C.c_int32, vp, vp,
                                          C.POINTER(Stats), vp]),
    "rt_render_tiles_device": (C.c_int32, [vp, C.POINTER(Pass), C.c_int32, C.c_int32, vp, vp,
                                           C.POINTER(Stats), vp]),
    "rt_render_accumulate": (C.c_int32, [vp, C.POINTER(Pass), C.c_int32, C.c_int32, C.c_int32,
                                         vp, vp, C.POINTER(Stats), vp]),
    "rt_render_tile_static": (C.c_int32, [vp, C.POINTER(Pass), C.c_int32, vp, vp, vp]),
    "rt_gather_probe": (C.c_int32, [C.c_int32, C.c_size_t, C.POINTER(C.c_double)]),
    "rt_resolve_device": (C.c_int32, [vp, C.c_size_t, C.c_int32, C.c_int32, C.c_float, vp, vp]),
}

_lib = None


def lib():
    """The loaded library; raises if it is not built."""
    global _lib
    if _lib is None:
        path = _build.LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m raytracer_amd.build` "
                "(hipcc, gfx950). There is no fallback implementation.")
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 and
        # loads it by path.  Importing torch first makes the loader resolve this
        # library's libamdhip64.so.7 dependency to that already-loaded copy
        # instead of mapping a second runtime from /opt/rocm (two runtimes in
        # one process see no devices).
        import torch  # noqa: F401
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise RtError(status, lib().rt_last_error().decode("utf-8", "replace"))
