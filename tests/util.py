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


# ---------------------------------------------------------------------------
# Ordered traversal (fast mode) against the reference order: every differing
# ray is put into a class, and anything outside the permitted classes fails.
# ---------------------------------------------------------------------------
def classify_fast_differences(O, verts, order, rays, fast, ref_tri, ref_t, degenerate_ok=False):
    """`fast`: HIT_DTYPE array of the ordered mode; ref_tri/ref_t: the reference
    order's hit positions and distances (oracle, or the bit-exact mode).
    Permitted classes for a ray whose hit id differs:
      closer_hit   the ordered mode reports a triangle that the oracle's own
                   triangle test (triangle.go:78-103 + the back-face cull of
                   bvh.go:121) accepts at 0 < t < t_ref with bit-identical
                   t, u, v: the reference's box test pruned it (its box test
                   is done in float32 on the exact boxes, the ordered mode's on
                   boxes that contain them)
      equal_t_earlier
                   the same with t == t_ref bit for bit, at an EARLIER array
                   position than the reference order's triangle: two triangles
                   share the hit point (an edge of the mesh) and the reference's
                   strict box test (boxTmin < *tmin, bvh.go:133-141) pruned the
                   box of the earlier one.  The opposite case -- the ordered mode
                   keeping the later triangle of such a tie -- is NOT permitted:
                   the ordered mode enters every box whose entry distance lies
                   within a few rounding steps of the hit distance, so it always
                   finds the earlier triangle.
    and, only with degenerate_ok (rays with a zero or denormal direction
    component, whose infinite inverse direction makes NaN slabs that the two
    slab forms treat differently):
      nan_slab     any other outcome on such a ray
    Returns the class counts."""
    verts = np.asarray(verts, np.float32).reshape(-1, 3, 3)
    counts = {"rays": int(len(rays)), "differing": 0, "closer_hit": 0, "equal_t_earlier": 0, "nan_slab": 0}
    differ = np.nonzero(fast["tri_pos"] != ref_tri)[0]
    counts["differing"] = int(len(differ))
    for i in differ:
        o, d = rays["origin"][i], rays["dir"][i]
        pf, pr = int(fast["tri_pos"][i]), int(ref_tri[i])
        degenerate = bool((np.abs(d) < np.float32(1.1754944e-38)).any())
        ok = False
        if pf >= 0:
            v = verts[order[pf]]
            n = np.zeros(3, np.float32)
            O.lib().orc_tri_normal(*[x.ctypes.data_as(C.c_void_p) for x in
                                     (np.ascontiguousarray(v[0]), np.ascontiguousarray(v[1]),
                                      np.ascontiguousarray(v[2]), n)])
            facing = not ((n[0] * d[0] + n[1] * d[1]) + n[2] * d[2] > 0)     # float32, in the reference's order
            t, u, w = O.tri_intersect(v[0], v[1], v[2], o, d)
            t32, u32, w32 = np.float32(t), np.float32(u), np.float32(w)
            same_numbers = (t32.view(np.uint32) == fast["t"][i].view(np.uint32)
                            and u32.view(np.uint32) == fast["u"][i].view(np.uint32)
                            and w32.view(np.uint32) == fast["v"][i].view(np.uint32))
            if facing and t32 > 0 and same_numbers:
                if t32 < ref_t[i]:
                    counts["closer_hit"] += 1
                    ok = True
                elif t32 == ref_t[i] and 0 <= pf < pr:
                    counts["equal_t_earlier"] += 1
                    ok = True
        if not ok:
            assert degenerate_ok and degenerate, \
                f"ray {i}: ordered mode reports triangle {pf} (t {fast['t'][i]!r}), reference order {pr} " \
                f"(t {ref_t[i]!r}): not a hit the reference's box test pruned"
            counts["nan_slab"] += 1
    return counts


def record_parity_counts(name, counts):
    """Merge one test's measured counts into gpurun_out/parity_counts.json (the
    file travels back from the GPU box; the round's copy is kept under
    profiles/)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "parity_counts.json")
    data = {}
    if os.path.exists(path):
        try:
            with open(path) as f:
                data = json.load(f)
        except ValueError:
            data = {}
    data[name] = counts
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(f"[parity counts] {name}: {counts}")
