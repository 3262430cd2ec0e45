"""Shared helpers for the tests: scene construction on both sides, Cornell box
loading, packing of rays."""
import ctypes as C
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def parse_obj_simple(obj_text, mtl_text):
    """Small OBJ/MTL reader for fixtures whose faces are all triangles.  Groups
    change when `usemtl` names a different material; Triangle.Index restarts
    per group (reference models/context.go:109-168)."""
    mats, names, cur = {}, [], None
    for line in mtl_text.splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "newmtl":
            cur = t[1]
            names.append(cur)
            mats[cur] = [1.0, 1.0, 1.0]
        elif t[0] == "Kd" and cur is not None:
            mats[cur] = [float(x) for x in t[1:4]]
    verts, tris, mat_ids, group_index = [], [], [], []
    cur_mtl, idx = None, 0
    for line in obj_text.splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            verts.append([float(x) for x in t[1:4]])
        elif t[0] == "usemtl":
            if t[1] != cur_mtl:
                cur_mtl, idx = t[1], 0
        elif t[0] == "f":
            ids = [int(p.split("/")[0]) - 1 for p in t[1:4]]
            tris.append(verts[ids[0]] + verts[ids[1]] + verts[ids[2]])
            mat_ids.append(names.index(cur_mtl))
            group_index.append(idx)
            idx += 1
    kd = np.array([mats[n] for n in names], np.float32)
    is_light = np.array([1 if n == "Light" else 0 for n in names], np.uint8)
    return {"verts": np.array(tris, np.float32), "material": np.array(mat_ids, np.int32),
            "group_index": np.array(group_index, np.int32), "kd": kd, "is_light": is_light,
            "names": names}


def cornell():
    d = os.path.join(GOLDEN, "cornell-box")
    with open(os.path.join(d, "cornell-box.obj")) as f:
        obj = f.read()
    with open(os.path.join(d, "cornell-box.mtl")) as f:
        mtl = f.read()
    return parse_obj_simple(obj, mtl)


def product_mesh(sc):
    from raytracer_amd import api
    return api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])


def oracle_scene(O, sc, bvh_export=None, build=None):
    """Oracle scene; the tree comes either from the product's export (so both
    sides walk the same tree) or from the oracle's own literal builder."""
    s = O.Scene(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    if bvh_export is not None:
        e = bvh_export
        s.set_bvh(e["is_leaf"], e["start"], e["end"], e["bmin"], e["bmax"], e["order"])
    elif build is not None:
        s.build_bvh(*build)
    return s


def rays6(rays):
    """(n,6) origin+direction and tmax from an rt_ray record array."""
    r = np.concatenate([rays["origin"], rays["dir"]], axis=1).astype(np.float32)
    return np.ascontiguousarray(r), np.ascontiguousarray(rays["tmax"])


def random_rays(n, seed, extent=1.5):
    """Incoherent rays: origins in a cube, normalised random directions."""
    from raytracer_amd import api, scenes
    u = scenes.splitmix_unit(seed, n * 6).reshape(n, 6)
    o = ((u[:, :3] * 2 - 1) * extent).astype(np.float32)
    d = (u[:, 3:] * 2 - 1)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
    return api.make_rays(o, d.astype(np.float32))


def struct_copy(src, cls):
    """Reinterpret one ctypes structure as another of identical layout."""
    assert C.sizeof(src) == C.sizeof(cls), (C.sizeof(src), C.sizeof(cls))
    dst = cls()
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(cls))
    return dst
