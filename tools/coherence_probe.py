#!/usr/bin/env python3
"""How much does the order of incoherent rays matter to the traversal kernels?

Traces the bench frame's primary rays, builds one diffuse bounce ray per hit
(random direction in the hemisphere of the hit triangle's normal), and times
the same set of bounce rays in several orders: pixel order (what the path
tracer's queue holds), grouped by direction octant within blocks of 512 rays,
and fully sorted by (octant, origin cell).

    python tools/coherence_probe.py [--mode fast|strict]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="fast")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    from raytracer_amd import api, scenes
    sc = scenes.room_mesh(706, seed=0x5EED0002, total=1_000_000)
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    bvh = api.Bvh.build(mesh, True, 8, 40)
    scene = api.Scene(mesh, bvh, 0)
    mode = {"strict": api.MODE_STRICT_F32, "fast": api.MODE_FAST_F32}[args.mode]
    o, d = scenes.primary_rays(args.width, args.height, sc["camera"])
    rays = api.make_rays(o, d)
    hits, st = scene.trace_rays(rays, mode=mode)
    print(f"primary: {len(rays)} rays, kernel {st['kernel_ms']:.2f} ms, {len(rays) / st['kernel_ms'] / 1e3:.0f} Mrays/s, "
          f"hit rate {(hits['tri_pos'] >= 0).mean():.3f}", flush=True)
    ok = hits["tri_pos"] >= 0
    order = bvh.export()["order"]
    v = sc["verts"].reshape(-1, 3, 3)[order][hits["tri_pos"][ok]]
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    flip = (n * rays["dir"][ok]).sum(axis=1) > 0
    n[flip] = -n[flip]
    rng = np.random.default_rng(3)
    nd = rng.normal(size=n.shape).astype(np.float32)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    below = (nd * n).sum(axis=1) < 0
    nd[below] = -nd[below]
    p = rays["origin"][ok] + rays["dir"][ok] * hits["t"][ok][:, None] + n.astype(np.float32) * np.float32(1e-3)
    b = api.make_rays(p.astype(np.float32), nd.astype(np.float32))
    octant = (nd[:, 0] < 0).astype(np.int64) | ((nd[:, 1] < 0).astype(np.int64) << 1) | ((nd[:, 2] < 0).astype(np.int64) << 2)
    lo, hi = p.min(axis=0), p.max(axis=0)
    cell = np.clip(((p - lo) / (hi - lo + 1e-6) * 64).astype(np.int64), 0, 63)
    cellkey = (cell[:, 0] << 12) | (cell[:, 1] << 6) | cell[:, 2]
    idx = np.arange(len(b))
    orders = {
        "pixel order": idx,
        "octant within blocks of 512": np.lexsort((octant, idx // 512)),
        "octant within blocks of 4096": np.lexsort((octant, idx // 4096)),
        "octant, then origin cell (global)": np.lexsort((cellkey, octant)),
        "origin cell, then octant (global)": np.lexsort((octant, cellkey)),
        "random shuffle": rng.permutation(len(b)),
    }
    for name, perm in orders.items():
        rb = np.ascontiguousarray(b[perm])
        best = 1e30
        for _ in range(3):
            h2, s2 = scene.trace_rays(rb, mode=mode)
            best = min(best, s2["kernel_ms"])
        print(f"{name:36s} {len(rb)} rays  kernel {best:8.2f} ms  {len(rb) / best / 1e3:7.0f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
