#!/usr/bin/env python3
"""BASELINE config[1] timed: 100 000-triangle soup (seed 0x5EED0001, SAH leaf 6,
depth 32), 1920x1080 primary rays through the pixel centres from (0, 0, 4),
cast as ONE batch through rt_trace_rays_device (rays and hits resident in HBM)
in every traversal mode, and by the CPU oracle's rayCast (process/trace.go:121-152
as restated) on 1 thread and on 16.

    python tools/config1_primary.py [OUT_JSON]        (needs a GPU)
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from oracle import oracle as O
    from raytracer_amd import api, scenes
    from util import oracle_scene, product_mesh, rays6
    sc = scenes.soup(100_000, seed=0x5EED0001)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 32)
    scene = api.Scene(mesh, bvh, 0)
    o, d = scenes.primary_rays(1920, 1080, {"x": 0.0, "y": 0.0, "z": 4.0})
    rays = api.make_rays(o, d)
    n = len(rays)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    gpu = {}
    for name, mode in (("strict_f32", api.MODE_STRICT_F32), ("fp64", api.MODE_FP64), ("fast_f32", api.MODE_FAST_F32),
                       ("fast_near_f32", api.MODE_FAST_NEAR_F32)):
        scene.trace_rays_device(d_rays.data_ptr(), n, d_hits.data_ptr(), mode=mode, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            scene.trace_rays_device(d_rays.data_ptr(), n, d_hits.data_ptr(), mode=mode, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        hits = d_hits.cpu().numpy()
        gpu[name] = {"ms_per_batch": round(ms, 4), "Mrays_per_s": round(n / ms / 1e3, 1),
                     "rays_hitting": int((hits[:, 0].view(np.int32) >= 0).sum())}
    st = scene.trace_rays(rays, mode=api.MODE_STRICT_F32, counters=True)[1]
    st_fast = scene.trace_rays(rays, mode=api.MODE_FAST_F32, counters=True)[1]

    osc = oracle_scene(O, sc, bvh_export=bvh.export())
    r6, tmax = rays6(rays)

    def cpu(n_threads, sample):
        idx = np.arange(0, n, max(1, n // sample))[:sample]
        parts = np.array_split(idx, n_threads)
        t0 = time.perf_counter()
        threads = [threading.Thread(target=lambda p=p: osc.trace_rays(np.ascontiguousarray(r6[p]), np.ascontiguousarray(tmax[p])))
                   for p in parts]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        return {"rays": int(len(idx)), "seconds": round(dt, 3), "Mrays_per_s": round(len(idx) / dt / 1e6, 3)}

    doc = {"workload": "cfg1: 100 000-triangle soup (seed 0x5EED0001), SAH leaf 6 depth 32, 1920x1080 primary rays, "
                       "tmax = MaxFloat32, one batch of rt_trace_rays_device with rays and hits in HBM",
           "rays": n, "gpu": gpu,
           "nodes_per_ray_reference_order": round(st["nodes_visited"] / n, 2),
           "tris_per_ray_reference_order": round(st["tris_fetched"] / n, 2),
           "bytes_per_ray_reference_order": round(48 + 56 * st["nodes_visited"] / n + 48 * st["tris_fetched"] / n, 1),
           "records_per_ray_ordered": round(st_fast["nodes_visited"] / n, 2),
           "tris_per_ray_ordered": round(st_fast["tris_fetched"] / n, 2),
           "bytes_per_ray_ordered": round(48 + 64 * st_fast["nodes_visited"] / n + 48 * st_fast["tris_fetched"] / n, 1),
           "cpu_oracle_1_thread": cpu(1, 400_000), "cpu_oracle_16_threads": cpu(16, n),
           "host_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
           "rt_version": api.version(), "git_head": os.environ.get("RT_GIT_HEAD") or None}
    text = json.dumps(doc, indent=1)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
