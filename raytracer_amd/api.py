"""Python host side over the C ABI: mesh, BVH and device scene objects.

Device memory, streams and torch.distributed come from PyTorch (plumbing);
every compute call goes through libraytracer_amd.so.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import Pass, Stats, check, lib

MODE_STRICT_F32 = 0
MODE_FP64 = 1
MODE_FAST_F32 = 2
MODE_FAST_NEAR_F32 = 3      # ordered traversal, rays starting within a scene diagonal of the scene's box

RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("tmax", np.float32),
                      ("dir", np.float32, 3), ("pad", np.float32)])
HIT_DTYPE = np.dtype([("tri_pos", np.int32), ("t", np.float32), ("u", np.float32),
                      ("v", np.float32)])
F32_MAX = np.float32(3.4028234663852886e38)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Mesh:
    """Triangles + materials: what RenderContext.Initialize produces
    (reference models/context.go:105-168)."""

    def __init__(self, verts, material=None, group_index=None, kd=None, is_light=None,
                 texcoords=None, texture=None):
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
        n = self.verts.shape[0]
        self.n_tris = n
        self.material = (np.zeros(n, np.int32) if material is None
                         else np.ascontiguousarray(material, np.int32))
        self.group_index = (np.arange(n, dtype=np.int32) if group_index is None
                            else np.ascontiguousarray(group_index, np.int32))
        self.kd = np.ascontiguousarray([[0.8, 0.8, 0.8]] if kd is None else kd,
                                       np.float32).reshape(-1, 3)
        m = self.kd.shape[0]
        self.is_light = (np.zeros(m, np.uint8) if is_light is None
                         else np.ascontiguousarray(is_light, np.uint8))
        self.texcoords = (None if texcoords is None
                          else np.ascontiguousarray(texcoords, np.float32).reshape(-1, 6))
        self.texture = (None if texture is None else np.ascontiguousarray(texture, np.int32))
        c = _capi.Mesh()
        c.verts = _ptr(self.verts)
        c.material = _ptr(self.material)
        c.group_index = _ptr(self.group_index)
        c.texcoords = _ptr(self.texcoords)
        c.n_tris = n
        c.kd = _ptr(self.kd)
        c.is_light = _ptr(self.is_light)
        c.texture = _ptr(self.texture)
        c.n_materials = m
        self.c = c

    def light(self, bvh=None):
        out = _capi.Light()
        check(lib().rt_mesh_light(C.byref(self.c), bvh._h if bvh is not None else None,
                                  C.byref(out)))
        return out


class Bvh:
    """Host BVH handle (reference models/bvh.go)."""

    def __init__(self, handle, mesh):
        self._h = handle
        self.mesh = mesh

    @classmethod
    def build(cls, mesh, use_bvh=True, max_leaf_size=6, max_depth=16):
        h = C.c_void_p()
        check(lib().rt_bvh_build(C.byref(mesh.c), int(use_bvh), max_leaf_size, max_depth,
                                 C.byref(h)))
        return cls(h, mesh)

    @classmethod
    def from_json(cls, mesh, text):
        raw = text.encode() if isinstance(text, str) else text
        h = C.c_void_p()
        check(lib().rt_bvh_from_json(C.byref(mesh.c), raw, len(raw), C.byref(h)))
        return cls(h, mesh)

    @classmethod
    def load(cls, mesh, path):
        h = C.c_void_p()
        check(lib().rt_bvh_load(C.byref(mesh.c), path.encode(), C.byref(h)))
        return cls(h, mesh)

    def save(self, path):
        check(lib().rt_bvh_save(self._h, path.encode()))

    def to_json(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib().rt_bvh_to_json(self._h, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value).decode()
        finally:
            lib().rt_free(p)

    @property
    def n_nodes(self):
        return lib().rt_bvh_node_count(self._h)

    @property
    def n_tris(self):
        return lib().rt_bvh_tri_count(self._h)

    @property
    def max_depth(self):
        return lib().rt_bvh_max_depth(self._h)

    def export(self):
        nn, nt = self.n_nodes, self.n_tris
        out = {
            "is_leaf": np.zeros(nn, np.int32), "start": np.zeros(nn, np.int32),
            "end": np.zeros(nn, np.int32), "depth": np.zeros(nn, np.int32),
            "bmin": np.zeros((nn, 3), np.float32), "bmax": np.zeros((nn, 3), np.float32),
            "split": np.zeros((nn, 4), np.float32), "order": np.zeros(nt, np.int32),
        }
        check(lib().rt_bvh_export(self._h, *[_ptr(out[k]) for k in
                                             ("is_leaf", "start", "end", "depth", "bmin", "bmax",
                                              "split", "order")]))
        return out

    def flatten(self):
        """(nodes (n_nodes,2,4) f32, tris (n_tris,3,4) f32, tri_material (n_tris,) i32)."""
        nodes = np.zeros((self.n_nodes, 2, 4), np.float32)
        tris = np.zeros((self.n_tris, 3, 4), np.float32)
        mat = np.zeros(self.n_tris, np.int32)
        check(lib().rt_flatten(C.byref(self.mesh.c), self._h, _ptr(nodes), _ptr(tris), _ptr(mat)))
        return nodes, tris, mat

    def flatten_ordered(self):
        """(records (n,16) f32, root_ref) of the ordered-traversal view."""
        n, root = C.c_size_t(), C.c_uint32()
        check(lib().rt_flatten_ordered(C.byref(self.mesh.c), self._h, None, 0, C.byref(n),
                                       C.byref(root)))
        rec = np.zeros((n.value, 16), np.float32)
        check(lib().rt_flatten_ordered(C.byref(self.mesh.c), self._h, _ptr(rec), n.value,
                                       C.byref(n), C.byref(root)))
        return rec, root.value

    def __del__(self):
        # at interpreter shutdown module globals may already be gone
        if getattr(self, "_h", None) and lib is not None:
            lib().rt_bvh_destroy(self._h)
            self._h = None


STATIC_WORKSPACE_BYTES = 64 << 20


def version():
    """The library's identification string (rt_version)."""
    return lib().rt_version().decode()


def workspace_bytes_per_path(light_sample_rays, bounce_limit):
    return int(lib().rt_workspace_bytes_per_path(light_sample_rays, bounce_limit))


def gather_probe(device, table_bytes):
    """Random 64-byte record gathers per second through a table of that size."""
    out = C.c_double()
    check(lib().rt_gather_probe(device, int(table_bytes), C.byref(out)))
    return out.value


def default_sample_table(n, seed):
    out = np.zeros((n, 2), np.float64)
    check(lib().rt_default_sample_table(n, seed, _ptr(out)))
    return out


def png_decode(data):
    """PNG bytes -> (h, w, 4) uint8 RGBA (premultiplied where the file has alpha)."""
    raw = bytes(data)
    w, h = C.c_int32(), C.c_int32()
    check(lib().rt_png_decode(raw, len(raw), None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    check(lib().rt_png_decode(raw, len(raw), _ptr(out), out.size, C.byref(w), C.byref(h)))
    return out


def camera_setup(cam, total_w, total_h):
    tl = np.zeros(3, np.float32)
    st = np.zeros(2, np.float32)
    check(lib().rt_camera_setup(C.byref(cam), total_w, total_h, _ptr(tl), _ptr(st)))
    return tl, st


def make_rays(origins, dirs, tmax=None):
    """Pack (n,3) origins and directions into the 32-byte rt_ray records."""
    o = np.asarray(origins, np.float32).reshape(-1, 3)
    d = np.asarray(dirs, np.float32).reshape(-1, 3)
    rays = np.zeros(o.shape[0], RAY_DTYPE)
    rays["origin"] = o
    rays["dir"] = d
    rays["tmax"] = F32_MAX if tmax is None else np.asarray(tmax, np.float32)
    return rays


class Scene:
    """Flattened scene resident on one GPU."""

    def __init__(self, mesh, bvh, device=0):
        h = C.c_void_p()
        check(lib().rt_scene_create(C.byref(mesh.c), bvh._h, device, C.byref(h)))
        self._h = h
        self.mesh = mesh
        self.bvh = bvh
        self.device = device

    def __del__(self):
        # at interpreter shutdown module globals may already be gone
        if getattr(self, "_h", None) and lib is not None:
            lib().rt_scene_destroy(self._h)
            self._h = None

    def info(self):
        out = _capi.SceneInfo()
        check(lib().rt_scene_info_get(self._h, C.byref(out)))
        vals = {n: getattr(out, n) for n, _ in out._fields_}
        return {n: (list(v) if isinstance(v, C.Array) else v) for n, v in vals.items()}

    def light(self):
        out = _capi.Light()
        check(lib().rt_scene_light(self._h, C.byref(out)))
        return out

    def set_sample_tables(self, camera_tab, light_tab):
        c = np.ascontiguousarray(camera_tab, np.float64).reshape(-1, 2)
        l = np.ascontiguousarray(light_tab, np.float64).reshape(-1, 2)
        check(lib().rt_scene_set_sample_tables(self._h, _ptr(c), c.shape[0], _ptr(l), l.shape[0]))

    def set_texture(self, texture_id, rgba):
        img = np.ascontiguousarray(rgba, np.uint8)
        h, w = img.shape[:2]
        check(lib().rt_scene_set_texture(self._h, texture_id, _ptr(img), w, h))

    def set_max_paths(self, n):
        check(lib().rt_scene_set_max_paths(self._h, n))

    def set_workspace_limit(self, max_bytes):
        """Bound on the device bytes of the tile renderer's workspace (0 = none)."""
        check(lib().rt_scene_set_workspace_limit(self._h, int(max_bytes)))

    def set_tuning(self, stack_entries=8, refill_below=0, blocks_per_cu=8, descend_min=0, leaf_min=0):
        """0 for a lane count = the kernel's own default."""
        check(lib().rt_scene_set_tuning(self._h, stack_entries, refill_below, blocks_per_cu, descend_min,
                                        leaf_min))

    def set_counters(self, enabled):
        check(lib().rt_scene_set_counters(self._h, int(enabled)))

    # ---- batch rayCast -------------------------------------------------
    def trace_rays(self, rays, mode=MODE_STRICT_F32, counters=False):
        """Host arrays in/out.  rays: RAY_DTYPE array.  Returns (hits, stats dict)."""
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(rays.shape[0], HIT_DTYPE)
        st = Stats()
        check(lib().rt_trace_rays(self._h, _ptr(rays), rays.shape[0], _ptr(hits), mode,
                                  int(counters), C.byref(st)))
        return hits, st.as_dict()

    def trace_rays_device(self, d_rays_ptr, n, d_hits_ptr, mode=MODE_STRICT_F32, stream=None,
                          d_counters_ptr=None):
        """Device pointers (e.g. torch tensor .data_ptr()); asynchronous."""
        if d_counters_ptr is None:
            check(lib().rt_trace_rays_device(self._h, d_rays_ptr, n, d_hits_ptr, mode, stream))
        else:
            check(lib().rt_trace_rays_device_counted(self._h, d_rays_ptr, n, d_hits_ptr, mode,
                                                     d_counters_ptr, stream))

    # ---- tile render ---------------------------------------------------
    def render_tile(self, p, mode=MODE_STRICT_F32):
        """Host buffers out: dict(rgba (h,w,4) u8, accum (h,w,3) f32, stats)."""
        w, h = p.width, p.height
        accum = np.zeros((h, w, 3), np.float32)
        rgba = np.zeros((h, w, 4), np.uint8)
        st = Stats()
        check(lib().rt_render_tile(self._h, C.byref(p), mode, _ptr(accum), _ptr(rgba),
                                   C.byref(st)))
        return {"rgba": rgba, "accum": accum, "stats": st.as_dict()}

    def render_tile_device(self, p, d_accum_ptr, d_rgba_ptr, mode=MODE_STRICT_F32, stream=None,
                           want_stats=False):
        st = Stats() if want_stats else None
        check(lib().rt_render_tile_device(self._h, C.byref(p), mode, d_accum_ptr, d_rgba_ptr,
                                          C.byref(st) if st is not None else None, stream))
        return st.as_dict() if st is not None else None

    def render_tile_static(self, p, d_accum_ptr, d_rgba_ptr, mode=MODE_STRICT_F32, stream=None):
        """One tile with the library's static device workspace: allocates nothing."""
        check(lib().rt_render_tile_static(self._h, C.byref(p), mode, d_accum_ptr, d_rgba_ptr, stream))

    def render_tiles_device(self, passes, d_accum_ptr, d_rgba_ptr, mode=MODE_STRICT_F32,
                            stream=None, want_stats=False):
        """Several tiles of one frame in one call; pixels of tile i follow those of
        tile i-1 in both device buffers.  `passes` is a list of Pass or a ready
        ctypes array of them (see pass_array)."""
        arr = passes if isinstance(passes, C.Array) else (Pass * len(passes))(*passes)
        st = Stats() if want_stats else None
        check(lib().rt_render_tiles_device(self._h, arr, len(passes), mode, d_accum_ptr,
                                           d_rgba_ptr, C.byref(st) if st is not None else None,
                                           stream))
        return st.as_dict() if st is not None else None

    def render_accumulate(self, p, sample_begin, sample_count, d_accum_ptr, d_rgba_ptr,
                          mode=MODE_STRICT_F32, stream=None, want_stats=False):
        st = Stats() if want_stats else None
        check(lib().rt_render_accumulate(self._h, C.byref(p), mode, sample_begin, sample_count,
                                         d_accum_ptr, d_rgba_ptr,
                                         C.byref(st) if st is not None else None, stream))
        return st.as_dict() if st is not None else None


def pass_array(passes):
    """ctypes array of Pass for render_tiles_device, reusable across calls (the
    caller may update fields such as rng_seed in place between frames)."""
    return (Pass * len(passes))(*passes)


def pass_from_dict(d):
    """RenderPass from the reference's JSON field names (models/context.go:40-52)."""
    p = Pass()
    ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    cam = d.get("Camera", {})
    tr = cam.get("Transform", ident)
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
    dl = s.get("DebugLightTransform", ident)
    for i in range(16):
        p.settings.debug_light_transform[i] = dl[i]
    return p
