#!/usr/bin/env python3
"""How evenly does the block-cyclic tile deal spread one frame's work over N
ranks?  Renders every tile of the bench frame on its own (device time and ray
count per tile), then prices the deal: max over ranks of the summed tile time /
mean.

    python tools/tile_balance.py [--workload cfg3] [--spp 4]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--spp", type=int, default=4)
    args = ap.parse_args()
    import torch
    import bench
    from raytracer_amd import api, scenes, tiles
    wl = bench.WORKLOADS[args.workload]
    sc = scenes.room_mesh(wl["grid"], seed=wl.get("seed", 0x5EED0002), total=wl["tris"])
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    bvh = api.Bvh.build(mesh, True, wl["leaf"], wl["depth"])
    scene = api.Scene(mesh, bvh, 0)
    tl = scenes.tiles(wl["width"], wl["height"], wl["tiles"])
    px = max(t[3] * t[4] for t in tl)
    d_accum = torch.empty((px * 3,), dtype=torch.float32, device="cuda")
    d_rgba = torch.zeros((px * 4,), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ms, rays = [], []
    for rep in range(2):
        ms, rays = [], []
        for t in tl:
            p = api.pass_from_dict(scenes.make_pass_dict(
                wl["width"], wl["height"], sc["camera"], spp=args.spp, bounces=wl["bounces"],
                light_samples=wl["light_samples"], intensity=wl["intensity"], seed=1 + t[0],
                x_offset=t[1], y_offset=t[2], tile_w=t[3], tile_h=t[4], task_id=t[0]))
            st = scene.render_tile_device(p, d_accum.data_ptr(), d_rgba.data_ptr(), mode=api.MODE_FAST_F32,
                                          stream=stream, want_stats=True)
            ms.append(st["total_ms"])
            rays.append(st["rays"])
    n = len(tl)
    out = {"workload": args.workload, "tiles": n, "spp": args.spp,
           "tile_ms_min_mean_max": [min(ms), sum(ms) / n, max(ms)],
           "tile_rays_min_mean_max": [min(rays), sum(rays) / n, max(rays)], "deals": {}}
    for world in (2, 4, 8):
        mine = [tiles.static_assignment(n, world, r) for r in range(world)]
        t_rank = [sum(ms[i] for i in m) for m in mine]
        r_rank = [sum(rays[i] for i in m) for m in mine]
        out["deals"][f"cyclic/{world}"] = {
            "time_max_over_mean": round(max(t_rank) / (sum(t_rank) / world), 4),
            "rays_max_over_mean": round(max(r_rank) / (sum(r_rank) / world), 4)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
