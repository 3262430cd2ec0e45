#!/usr/bin/env python3
"""Times the traversal kernels on representative ray sets of the 1M-triangle
room (primary rays, bounce rays from the primary hit points) for a sweep of
launch parameters.  GPU only; for development, not part of the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracer_amd import api, scenes  # noqa: E402


def main():
    grid = int(os.environ.get("GRID", "706"))
    total = int(os.environ.get("TRIS", "1000000"))
    sc = scenes.room_mesh(grid, seed=0x5EED0002, total=total)
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    bvh = api.Bvh.build(mesh, True, 8, 40)
    scene = api.Scene(mesh, bvh, 0)
    w, h = 2560, 1440
    o, d = scenes.primary_rays(w, h, sc["camera"])
    prim = api.make_rays(o, d)
    hits, _ = scene.trace_rays(prim, mode=api.MODE_FAST_F32)
    ok = hits["tri_pos"] >= 0
    pts = prim["origin"][ok] + prim["dir"][ok] * hits["t"][ok][:, None]
    n = pts.shape[0]
    u = scenes.splitmix_unit(7, n * 3).reshape(n, 3) * 2 - 1
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    u[:, 1] = np.abs(u[:, 1])          # upper hemisphere (the field faces +y)
    bounce = api.make_rays(pts.astype(np.float32), u.astype(np.float32))
    sets = {"primary": prim, "bounce": bounce}
    stream = torch.cuda.current_stream().cuda_stream
    results = []
    for name, rays in sets.items():
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
        d_hits = torch.zeros((len(rays), 4), dtype=torch.float32, device="cuda")

        def run(mode, reps=5):
            scene.trace_rays_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), mode=mode, stream=stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                scene.trace_rays_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), mode=mode, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            return len(rays) * reps / (e0.elapsed_time(e1) * 1e-3) / 1e6

        scene.set_tuning(16, 40, 8)
        print(f"{name:8s} n={len(rays)} strict {run(api.MODE_STRICT_F32):8.1f} Mrays/s", flush=True)
        for stack in (8, 12, 16, 24):
            for refill in (16, 32, 40, 56):
                for bpc in (5, 8):
                    scene.set_tuning(stack, refill, bpc)
                    r = run(api.MODE_FAST_F32)
                    results.append((name, stack, refill, bpc, r))
                    print(f"{name:8s} stack {stack:2d} refill {refill:2d} blocks/cu {bpc}: {r:8.1f} Mrays/s", flush=True)
    print("done")


if __name__ == "__main__":
    main()
