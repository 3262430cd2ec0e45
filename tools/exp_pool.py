#!/usr/bin/env python3
"""Experiment driver: the ordered traversal with different RT_POOL settings
(read by the library when a scene is created); checks that hits do not change
and prints traversal / frame rates.   python tools/exp_pool.py [--spp 8] "" "1,24,48,16" ..."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--small", action="store_true", help="only the 400k-ray probe (first contact with a new kernel)")
    ap.add_argument("--tune", default="")
    ap.add_argument("sets", nargs="*", default=[""])
    args = ap.parse_args()
    import torch
    import bench
    from raytracer_amd import api, scenes
    from util import random_rays
    wl = bench.WORKLOADS["small" if args.small else args.workload]
    sc = scenes.room_mesh(wl["grid"], seed=wl.get("seed", 0x5EED0002), total=wl["tris"])
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    bvh = api.Bvh.build(mesh, True, wl["leaf"], wl["depth"])
    tl = scenes.tiles(wl["width"], wl["height"], wl["tiles"])
    px = sum(t[3] * t[4] for t in tl)
    d_accum = torch.empty((px * 3,), dtype=torch.float32, device="cuda")
    d_rgba = torch.zeros((px * 4,), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    passes = api.pass_array([api.pass_from_dict(scenes.make_pass_dict(
        wl["width"], wl["height"], sc["camera"], spp=args.spp, bounces=wl["bounces"],
        light_samples=wl["light_samples"], intensity=wl["intensity"], seed=1 + t[0],
        x_offset=t[1], y_offset=t[2], tile_w=t[3], tile_h=t[4], task_id=t[0])) for t in tl])
    half = wl["grid"] * 0.05 / 2.0
    probe = random_rays(400_000, seed=5, extent=1.0)
    probe["origin"] = probe["origin"] * np.float32([0.9 * half, 0.1 * half, 0.9 * half]) + np.float32([0, 0.6 * half, 0])
    probe["tmax"][::7] = np.float32(half)
    base_hits = base_img = base_cnt = None
    for spec in args.sets:
        if spec:
            os.environ["RT_POOL"] = spec
        else:
            os.environ.pop("RT_POOL", None)
        scene = api.Scene(mesh, bvh, 0)
        if args.tune:
            scene.set_tuning(*[int(x) for x in args.tune.split(",")])
        print(f"[{spec}] probe ...", flush=True)
        hits, st = scene.trace_rays(probe, mode=api.MODE_FAST_F32, counters=True)
        cnt = (st["nodes_visited"], st["box_tests"], st["tris_fetched"], st["tris_tested"])
        hits2, _ = scene.trace_rays(probe, mode=api.MODE_FAST_F32)
        if base_hits is None:
            base_hits, base_cnt = hits, cnt
        same = np.array_equal(hits.view(np.uint32), base_hits.view(np.uint32)) and \
            np.array_equal(hits2.view(np.uint32), base_hits.view(np.uint32))
        print(f"[{spec}] probe hits_same={same} counters_same={cnt == base_cnt}", flush=True)
        if args.small:
            continue
        best = None
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s = scene.render_tiles_device(passes, d_accum.data_ptr(), d_rgba.data_ptr(), mode=api.MODE_FAST_F32,
                                          stream=stream, want_stats=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, s)
        img = d_rgba.cpu().numpy()
        if base_img is None:
            base_img = img
        dt, s = best
        print(f"[{spec:18s}] image_same={np.array_equal(img, base_img)} frame {s['rays'] / dt / 1e6:8.1f} Mrays/s  "
              f"traversal {s['rays'] / s['kernel_ms'] / 1e3:8.1f} Mrays/s ({s['kernel_ms']:.1f} ms of {dt * 1e3:.1f})",
              flush=True)
        del scene


if __name__ == "__main__":
    main()
