#!/usr/bin/env python3
"""CPU study of the ordered traversal's step (see ordering_study.cpp): records
visited and triangles fetched per ray for the kernel's policy and for the
sort-free variants, on the bench scene (1M-triangle room) with three ray sets:
primary rays of the bench camera, diffuse bounce rays leaving their hit points,
and shadow rays from those points to the light.

    python tools/study/ordering_study.py [n_rays] > profiles/r03_ordering_study.json
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracer_amd import api, scenes  # noqa: E402


def build():
    src = os.path.join(ROOT, "tools", "study", "ordering_study.cpp")
    out = os.path.join(ROOT, "tools", "study", "libordering_study.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DRT_HD=inline",
                               "-o", out, src])
    return C.CDLL(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    lib = build()
    sc = scenes.room_mesh(706, seed=0x5EED0002, total=1_000_000)
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    bvh = api.Bvh.build(mesh, True, 8, 40)
    rec, root = bvh.flatten_ordered()
    _nodes, tris, _mat = bvh.flatten()
    rec = np.ascontiguousarray(rec, np.float32)
    tris = np.ascontiguousarray(tris, np.float32)

    def run(rays, policy, visits=None):
        r8 = np.ascontiguousarray(rays.view(np.float32).reshape(-1, 8))
        t = np.zeros(len(rays), np.float32)
        pos = np.zeros(len(rays), np.int32)
        out = np.zeros(3, np.uint64)
        lib.ordering_study(rec.ctypes.data_as(C.c_void_p), C.c_uint32(root), tris.ctypes.data_as(C.c_void_p),
                           C.c_int32(tris.shape[0]), r8.ctypes.data_as(C.c_void_p), C.c_int64(len(rays)),
                           C.c_int(policy), t.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                           out.ctypes.data_as(C.c_void_p), visits.ctypes.data_as(C.c_void_p) if visits is not None else None)
        return t, pos, out

    # ray sets: primary rays of the bench camera (a strided sample of the 4K frame), then
    # from their hit points a uniform hemisphere bounce ray and a shadow ray to a light point
    o, d = scenes.primary_rays(3840, 2160, sc["camera"])
    pick = np.random.default_rng(1).choice(len(o), n, replace=False)
    prim = api.make_rays(o[pick], d[pick])
    t, pos, _ = run(prim, 0)
    ok = pos >= 0
    pts = (prim["origin"][ok] + prim["dir"][ok] * t[ok][:, None]).astype(np.float32)
    order = bvh.export()["order"]
    v = sc["verts"].reshape(-1, 3, 3)[order][pos[ok]]
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    rng = np.random.default_rng(2)
    u = rng.normal(size=pts.shape).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    u[(u * nrm).sum(axis=1) < 0] *= -1
    bounce = api.make_rays(pts, u.astype(np.float32))
    light_tris = sc["verts"].reshape(-1, 3, 3)[np.asarray(sc["is_light"])[sc["material"]] == 1]
    lc = light_tris.reshape(-1, 3).mean(axis=0)
    to = lc[None, :] - pts
    dist = np.linalg.norm(to, axis=1).astype(np.float32)
    shadow = api.make_rays(pts, (to / dist[:, None]).astype(np.float32), tmax=dist)

    doc = {"scene": "1M-triangle room (bench workload cfg3), SAH leaf 8, ordered view of %d records" % len(rec),
           "rays_per_set": {"primary": int(len(prim)), "bounce": int(len(bounce)), "shadow": int(len(shadow))},
           "policies": {"0": "kernel today: children sorted by entry distance, one (child, distance) stack entry per child",
                        "1": "octant order (front to back along the ray octant's diagonal) + ONE entry per record "
                             "(children left, smallest entry distance); popped children entered untested",
                        "2": "distance order + one entry per record",
                        "3": "octant order + one (child, distance) entry per child"},
           "results": {}}
    for name, rays in (("primary", prim), ("bounce", bounce), ("shadow", shadow)):
        base_t = None
        for policy in (0, 1, 2, 3):
            t, pos, out = run(rays, policy)
            if policy == 0:
                base_t = t
            doc["results"].setdefault(name, {})[str(policy)] = {
                "records_per_ray": round(float(out[0]) / len(rays), 3),
                "triangles_per_ray": round(float(out[1]) / len(rays), 3),
                "stale_children_entered_per_ray": round(float(out[2]) / len(rays), 3),
                "hit_distance_differs_from_policy_0": int((t.view(np.uint32) != base_t.view(np.uint32)).sum())}
    # Which share of a ray's record fetches would a table of K records in LDS serve?  Three ways to
    # choose the K: the most visited (the best any choice can do), the largest frames (surface area of
    # the 255-step box, known at build time), breadth-first from the root.
    refs = rec.view(np.uint32)[:, 12:16]
    step = rec[:, 3:6].astype(np.float64) * 255.0
    area = step[:, 0] * step[:, 1] + step[:, 0] * step[:, 2] + step[:, 1] * step[:, 2]
    by_area = np.argsort(-area, kind="stable")
    bfs, frontier = [], [int(root)]
    while frontier and len(bfs) < 8192:
        bfs.extend(frontier)
        kids = refs[frontier].ravel()
        frontier = [int(k) for k in kids if k < 0x80000000]
    by_bfs = np.array(bfs[:8192])
    doc["top_table_share"] = {"K": [64, 128, 256, 512, 1024, 2048, 4096]}
    for name, rays in (("primary", prim), ("bounce", bounce), ("shadow", shadow)):
        visits = np.zeros(len(rec), np.uint32)
        _t, _pos, out = run(rays, 4, visits)
        total = float(visits.sum())
        by_visits = np.argsort(-visits.astype(np.int64), kind="stable")
        doc["top_table_share"][name] = {
            "records_per_ray": round(total / len(rays), 3),
            "most_visited": [round(float(visits[by_visits[:k]].sum()) / total, 3) for k in doc["top_table_share"]["K"]],
            "largest_frames": [round(float(visits[by_area[:k]].sum()) / total, 3) for k in doc["top_table_share"]["K"]],
            "breadth_first": [round(float(visits[by_bfs[:k]].sum()) / total, 3) for k in doc["top_table_share"]["K"]]}
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
