"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()``
and by the ``cpu_baseline`` leg of ``bench.py`` as the checker / CPU baseline.
The GPU product path never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

MODE_F32 = 0
MODE_F64 = 1


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("rt_oracle.c", "rt_walk.inc", "rt_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays", "nodes_visited", "box_tests", "tris_fetched", "tris_tested")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


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


class Light(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("size", C.c_float * 2),
                ("emission", C.c_float * 3), ("normal", C.c_float * 3), ("found", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, vp, vp, C.c_int32, vp, vp, C.c_int32]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_set_texcoords.argtypes = [vp, vp]
        L.orc_scene_set_texture.restype = C.c_int32
        L.orc_scene_set_texture.argtypes = [vp, C.c_int32, vp, C.c_int32, C.c_int32]
        L.orc_build_bvh.restype = C.c_int32
        L.orc_build_bvh.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
        L.orc_set_bvh.restype = C.c_int32
        L.orc_set_bvh.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
        L.orc_node_count.restype = C.c_int32
        L.orc_node_count.argtypes = [vp]
        L.orc_export_bvh.argtypes = [vp] * 9
        L.orc_scene_light.argtypes = [vp, C.POINTER(Light)]
        L.orc_set_sample_tables.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
        L.orc_trace_rays.argtypes = [vp, vp, vp, C.c_int64, C.c_int, vp, vp, vp, vp,
                                     C.POINTER(Counters)]
        L.orc_camera_setup.argtypes = [C.POINTER(Camera), C.c_int32, C.c_int32, vp, vp]
        L.orc_camera_ray.argtypes = [vp, C.POINTER(Pass), C.c_int32, C.c_int32, C.c_uint64, vp]
        L.orc_render_tile.argtypes = [vp, C.POINTER(Pass), C.c_int32, C.c_int32, C.c_int32,
                                      vp, vp, vp, C.POINTER(Counters)]
        L.orc_render_progress.restype = C.c_int32
        L.orc_render_progress.argtypes = [vp, C.POINTER(Pass), C.c_uint64, vp, vp, C.c_int32]
        L.orc_set_light_index_mode.argtypes = [vp, C.c_int32]
        L.orc_resolve.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, vp]
        L.orc_aabb_intersect.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_tri_intersect.argtypes = [vp, vp, vp, vp, vp, C.c_int] + [C.POINTER(C.c_double)] * 3
        L.orc_tri_normal.argtypes = [vp, vp, vp, vp]
        for f in ("orc_go_cos", "orc_go_exp", "orc_go_log"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double]
        for f in ("orc_go_pow", "orc_go_min", "orc_go_max"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double, C.c_double]
        L.orc_rng_stream.restype = C.c_uint64
        L.orc_rng_stream.argtypes = [C.c_int64, C.c_uint64]
        L.orc_rng_f32.restype = C.c_float
        L.orc_rng_f32.argtypes = [C.c_uint64, C.c_uint32]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def aabb_intersect(bmin, bmax, o, d, mode=MODE_F32):
    hit, t0, t1 = C.c_int(), C.c_double(), C.c_double()
    a, b, oo, dd = _f32(bmin), _f32(bmax), _f32(o), _f32(d)
    lib().orc_aabb_intersect(_ptr(a), _ptr(b), _ptr(oo), _ptr(dd), mode,
                             C.byref(hit), C.byref(t0), C.byref(t1))
    return bool(hit.value), t0.value, t1.value


def tri_intersect(v0, v1, v2, o, d, mode=MODE_F32):
    t, u, v = C.c_double(), C.c_double(), C.c_double()
    arrs = [_f32(x) for x in (v0, v1, v2, o, d)]
    lib().orc_tri_intersect(*[_ptr(x) for x in arrs], mode, C.byref(t), C.byref(u), C.byref(v))
    return t.value, u.value, v.value


def pass_from_dict(d, cls=Pass):
    """Fill a pass structure from the reference's RenderPass JSON field names
    (models/context.go:40-52, camera.go:19-31, settings.go:5-16)."""
    p = cls()
    cam = d.get("Camera", {})
    tr = cam.get("Transform", [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1])
    for i in range(16):
        p.camera.transform[i] = tr[i]
    p.camera.projection_plane_distance = cam.get("ProjectionPlaneDistance", 1.0)
    p.camera.rays_per_pixel = cam.get("RaysPerPixel", 1)
    p.camera.projection = cam.get("Projection", 0)
    p.camera.field_of_view = cam.get("FieldOfView", 45.0)
    p.camera.ortographic_size = cam.get("OrtographicSize", 3.0)
    p.total_width = d.get("TotalWidth", d.get("Width", 0))
    p.total_height = d.get("TotalHeight", d.get("Height", 0))
    p.task_id = d.get("TaskID", 0)
    p.rng_seed = d.get("RNGSeed", 0)
    p.x_offset = d.get("XOffset", 0)
    p.y_offset = d.get("YOffset", 0)
    p.width = d.get("Width", p.total_width)
    p.height = d.get("Height", p.total_height)
    p.render_key = d.get("RenderKey", 0)
    s = d.get("Settings", {})
    p.settings.draw_surface_normal = int(s.get("DrawSurfaceNormal", False))
    p.settings.gamma_correction = int(s.get("GammaCorrection", False))
    p.settings.gamma = s.get("Gamma", 2.2)
    p.settings.light_sample_rays = s.get("LightSampleRays", 1)
    p.settings.bounce_limit = s.get("BounceLimit", 0)
    p.settings.light_intensity = s.get("LightIntensity", 100.0)
    p.settings.debug_light_size = s.get("DebugLightSize", 1.0)
    p.settings.force_debug_light = int(s.get("ForceDebugLight", False))
    p.settings.debug_light_at_camera = int(s.get("DebugLightAtCamera", False))
    dl = s.get("DebugLightTransform", [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1])
    for i in range(16):
        p.settings.debug_light_transform[i] = dl[i]
    return p


class Scene:
    """Oracle scene: triangles + (after build_bvh / set_bvh) the pointer tree."""

    def __init__(self, verts, material=None, group_index=None, kd=None, is_light=None):
        self.verts = _f32(verts).reshape(-1, 9)
        n = self.verts.shape[0]
        self.n = n
        self.material = (np.zeros(n, np.int32) if material is None
                         else np.ascontiguousarray(material, np.int32))
        self.group_index = (np.arange(n, dtype=np.int32) if group_index is None
                            else np.ascontiguousarray(group_index, np.int32))
        self.kd = _f32([[0.8, 0.8, 0.8]] if kd is None else kd).reshape(-1, 3)
        m = self.kd.shape[0]
        self.is_light = (np.zeros(m, np.uint8) if is_light is None
                         else np.ascontiguousarray(is_light, np.uint8))
        self._h = lib().orc_scene_create(_ptr(self.verts), _ptr(self.material),
                                         _ptr(self.group_index), n, _ptr(self.kd),
                                         _ptr(self.is_light), m)

    def __del__(self):
        # at interpreter shutdown module globals may already be gone
        if getattr(self, "_h", None) and lib is not None:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def set_texcoords(self, uv):
        """Per-vertex texture coordinates, (n, 6) in input order (context.go:138-156)."""
        uv = _f32(uv).reshape(-1, 6)
        assert uv.shape[0] == self.n
        lib().orc_scene_set_texcoords(self._h, _ptr(uv))

    def set_texture(self, material, rgba):
        """The image behind one material's map_Kd: (h, w, 4) uint8."""
        img = np.ascontiguousarray(rgba, np.uint8)
        h, w = img.shape[:2]
        if lib().orc_scene_set_texture(self._h, int(material), _ptr(img), w, h) != 0:
            raise ValueError("bad texture arguments")

    def build_bvh(self, use_bvh=True, max_leaf=6, max_depth=16):
        return lib().orc_build_bvh(self._h, int(use_bvh), max_leaf, max_depth)

    def set_bvh(self, is_leaf, start, end, bmin, bmax, order):
        a = [np.ascontiguousarray(x, np.int32) for x in (is_leaf, start, end)]
        b = [_f32(bmin), _f32(bmax)]
        o = np.ascontiguousarray(order, np.int32)
        return lib().orc_set_bvh(self._h, len(a[0]), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]),
                                 _ptr(b[0]), _ptr(b[1]), _ptr(o))

    def export_bvh(self):
        nn = lib().orc_node_count(self._h)
        out = {
            "is_leaf": np.zeros(nn, np.int32), "start": np.zeros(nn, np.int32),
            "end": np.zeros(nn, np.int32), "depth": np.zeros(nn, np.int32),
            "bmin": np.zeros((nn, 3), np.float32), "bmax": np.zeros((nn, 3), np.float32),
            "split": np.zeros((nn, 4), np.float32), "order": np.zeros(self.n, np.int32),
        }
        lib().orc_export_bvh(self._h, *[_ptr(out[k]) for k in
                                        ("is_leaf", "start", "end", "depth", "bmin", "bmax",
                                         "split", "order")])
        return out

    def light(self):
        lt = Light()
        lib().orc_scene_light(self._h, C.byref(lt))
        return lt

    def set_sample_tables(self, camera_tab, light_tab):
        c = np.ascontiguousarray(camera_tab, np.float64).reshape(-1, 2)
        l = np.ascontiguousarray(light_tab, np.float64).reshape(-1, 2)
        lib().orc_set_sample_tables(self._h, _ptr(c), c.shape[0], _ptr(l), l.shape[0])

    def set_light_index_mode(self, mode):
        """0: hashed row per camera sample; 1: the reference's advancing index per light."""
        lib().orc_set_light_index_mode(self._h, int(mode))

    def trace_rays(self, rays, tmax=None, mode=MODE_F32, counters=False):
        """rays: (n,6) origin+direction.  Returns dict(tri, t, u, v[, counters])."""
        r = _f32(rays).reshape(-1, 6)
        n = r.shape[0]
        tm = None if tmax is None else _f32(tmax)
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        cnt = Counters()
        lib().orc_trace_rays(self._h, _ptr(r), _ptr(tm), n, mode, _ptr(tri), _ptr(t), _ptr(u),
                             _ptr(v), C.byref(cnt) if counters else None)
        out = {"tri": tri, "t": t, "u": u, "v": v}
        if counters:
            out["counters"] = cnt.as_dict()
        return out

    def camera_ray(self, p, x, y, path_index):
        ray = np.zeros(6, np.float32)
        lib().orc_camera_ray(self._h, C.byref(p), x, y, path_index, _ptr(ray))
        return ray

    def render_tile(self, p, sample_begin=0, sample_count=None, incremental=False,
                    accum=None, want_radiance=False, counters=True):
        """main.go render(): returns dict(rgba (h,w,4) u8, accum (h,w,3) f32, ...)."""
        if sample_count is None:
            sample_count = p.camera.rays_per_pixel
        w, h = p.width, p.height
        acc = np.zeros((h, w, 3), np.float32) if accum is None else accum
        rgba = np.zeros((h, w, 4), np.uint8)
        rad = np.zeros((sample_count, h, w, 3), np.float32) if want_radiance else None
        cnt = Counters()
        lib().orc_render_tile(self._h, C.byref(p), sample_begin, sample_count,
                              int(incremental), _ptr(acc), _ptr(rgba), _ptr(rad),
                              C.byref(cnt) if counters else None)
        out = {"rgba": rgba, "accum": acc, "counters": cnt.as_dict()}
        if want_radiance:
            out["radiance"] = rad
        return out


def render_progress(scene, p, base_rays=0, max_events=64):
    """[(progress, rays)] of the "trace" events render() issues (main.go:120-154)."""
    prog = np.zeros(max_events, np.float32)
    rays = np.zeros(max_events, np.uint64)
    n = lib().orc_render_progress(scene._h, C.byref(p), int(base_rays), _ptr(prog), _ptr(rays), max_events)
    return [(float(prog[i]), int(rays[i])) for i in range(min(n, max_events))]


def resolve(accum, divisor, gamma_on, gamma):
    a = _f32(accum).reshape(-1, 3)
    out = np.zeros((a.shape[0], 4), np.uint8)
    lib().orc_resolve(_ptr(a), a.shape[0], divisor, int(gamma_on), gamma, _ptr(out))
    return out


def camera_setup(cam, total_w, total_h):
    tl = np.zeros(3, np.float32)
    st = np.zeros(2, np.float32)
    lib().orc_camera_setup(C.byref(cam), total_w, total_h, _ptr(tl), _ptr(st))
    return tl, st
