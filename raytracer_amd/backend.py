"""Host-side mirror of the reference backend's entry points (reference
src/backend/main.go:28-33): same function names, same JSON-string arguments and
results, one Backend object per worker / GPU instead of package globals.

    b = Backend(device=0, progressUpdate=print)
    b.initialize(json.dumps({"UseBVH": True, "BVHMaxLeafSize": 6, "BVHMaxDepth": 16,
                             "ObjBuffer": obj_text, "MtlBuffer": mtl_text, "RawTextures": []}))
    b.loadBVH(b.buildBVH())
    result = json.loads(b.render(json.dumps(render_pass)))      # exitCode, message, imageData
"""
import base64
import ctypes as C
import json

import numpy as np

from . import _capi
from ._capi import check, vp

_PROGRESS_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)

_SIGNATURES = {
    "rtb_create": (C.c_int32, [C.c_int32, C.POINTER(vp)]),
    "rtb_destroy": (None, [vp]),
    "rtb_set_progress_callback": (None, [vp, _PROGRESS_FN, vp]),
    "rtb_set_mode": (C.c_int32, [vp, C.c_int32]),
    "rtb_initialize": (C.c_int32, [vp, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t),
                                   C.c_int32]),
    "rtb_build_bvh": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "rtb_load_bvh": (C.c_int32, [vp, C.c_char_p, C.c_size_t]),
    "rtb_render": (C.c_int32, [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "rtb_initialize_incremental_render": (C.c_int32, [vp, C.c_char_p, C.POINTER(vp),
                                                      C.POINTER(C.c_size_t)]),
    "rtb_incremental_render": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "rtb_rays": (C.c_uint64, [vp]),
    "rtb_progress_deliveries": (C.c_uint32, [vp]),
    "rtb_triangle_count": (C.c_int32, [vp]),
    "rtb_last_image": (C.c_size_t, [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}
_bound = False


def _lib():
    global _bound
    L = _capi.lib()
    if not _bound:
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return L


def decode_image(result):
    """(h, w, 4) uint8 array from a result document (dict or JSON text)."""
    if isinstance(result, (str, bytes)):
        result = json.loads(result)
    img = result["imageData"]
    w, h = img["Rect"]["Max"]["X"], img["Rect"]["Max"]["Y"]
    return np.frombuffer(base64.b64decode(img["Pix"]), np.uint8).reshape(h, w, 4)


class Backend:
    def __init__(self, device=0, progressUpdate=None, mode=None):
        h = vp()
        check(_lib().rtb_create(device, C.byref(h)))
        self._h = h
        self._cb = None
        if progressUpdate is not None:
            self._cb = _PROGRESS_FN(lambda text, _user: progressUpdate(text.decode()))
            _lib().rtb_set_progress_callback(self._h, self._cb, None)
        if mode is not None:
            check(_lib().rtb_set_mode(self._h, mode))

    def set_mode(self, mode):
        """Traversal mode of the render calls (api.MODE_*)."""
        check(_lib().rtb_set_mode(self._h, mode))

    def __del__(self):
        # at interpreter shutdown module globals may already be gone
        if getattr(self, "_h", None) and _lib is not None:
            _lib().rtb_destroy(self._h)
            self._h = None

    def _take(self, p, n):
        if not p:
            return None
        try:
            return C.string_at(p, n.value).decode()
        finally:
            _lib().rt_free(p)

    # -- the six entry points, names as in main.go:28-33 ----------------------
    def initialize(self, context_json, *textures):
        arr = (C.c_char_p * max(len(textures), 1))(*[bytes(t) for t in textures])
        sizes = (C.c_size_t * max(len(textures), 1))(*[len(t) for t in textures])
        check(_lib().rtb_initialize(self._h, context_json.encode(), arr, sizes, len(textures)))

    def buildBVH(self):
        p, n = vp(), C.c_size_t()
        check(_lib().rtb_build_bvh(self._h, C.byref(p), C.byref(n)))
        return self._take(p, n)

    def loadBVH(self, bvh_json):
        raw = bvh_json.encode()
        check(_lib().rtb_load_bvh(self._h, raw, len(raw)))

    def render(self, pass_json):
        """Returns the result document; failures come back as exitCode -1 like
        the reference's handleError, never as an exception."""
        p, n = vp(), C.c_size_t()
        _lib().rtb_render(self._h, pass_json.encode(), C.byref(p), C.byref(n))
        return self._take(p, n)

    def initializeIncrementalRender(self, pass_json):
        p, n = vp(), C.c_size_t()
        _lib().rtb_initialize_incremental_render(self._h, pass_json.encode(), C.byref(p), C.byref(n))
        return self._take(p, n)

    def incrementalRender(self, pass_json=None):
        p, n = vp(), C.c_size_t()
        _lib().rtb_incremental_render(self._h, C.byref(p), C.byref(n))
        return self._take(p, n)

    # -- extras ---------------------------------------------------------------
    @property
    def rays(self):
        return int(_lib().rtb_rays(self._h))

    @property
    def progress_deliveries(self):
        """Separate moments at which the last render() sent its trace updates."""
        return int(_lib().rtb_progress_deliveries(self._h))

    @property
    def triangle_count(self):
        return int(_lib().rtb_triangle_count(self._h))

    def last_image(self):
        p, w, h = vp(), C.c_int32(), C.c_int32()
        n = _lib().rtb_last_image(self._h, C.byref(p), C.byref(w), C.byref(h))
        if n == 0:
            return None
        return np.frombuffer(C.string_at(p, n), np.uint8).reshape(h.value, w.value, 4).copy()
