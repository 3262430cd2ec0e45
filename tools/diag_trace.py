#!/usr/bin/env python3
"""Lane-utilisation diagnostics of the ordered traversal kernel on the bench
frame: renders the whole frame at --spp with the counting kernel and prints the
wave-level step counts next to the per-lane ones (needs RT_DIAG=1 for the
library's own stderr line).

    RT_DIAG=1 python tools/diag_trace.py --spp 1
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--grid", type=int, default=706)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--leaf", type=int, default=8)
    ap.add_argument("--mode", default="fast")
    ap.add_argument("--tune", default="", help="stack,refill,blocks_per_cu,descend_min,leaf_min[;...]: several = sweep")
    args = ap.parse_args()
    import torch
    from raytracer_amd import api, scenes
    sc = scenes.room_mesh(args.grid, seed=0x5EED0002, total=args.tris)
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    bvh = api.Bvh.build(mesh, True, args.leaf, 40)
    scene = api.Scene(mesh, bvh, 0)
    tunes = [t for t in args.tune.split(";") if t]
    if len(tunes) == 1:
        scene.set_tuning(*[int(x) for x in tunes[0].split(",")])
    W, H = 3840, 2160
    tl = scenes.tiles(W, H, 16)
    px = sum(t[3] * t[4] for t in tl)
    d_accum = torch.empty((px * 3,), dtype=torch.float32, device="cuda")
    d_rgba = torch.zeros((px * 4,), dtype=torch.uint8, device="cuda")
    mode = {"strict": api.MODE_STRICT_F32, "fast": api.MODE_FAST_F32}[args.mode]
    passes = [api.pass_from_dict(scenes.make_pass_dict(
        W, H, sc["camera"], spp=args.spp, bounces=4, light_samples=1, intensity=6.0, seed=1,
        x_offset=x, y_offset=y, tile_w=w, tile_h=h, task_id=tid)) for (tid, x, y, w, h) in tl]
    stream = torch.cuda.current_stream().cuda_stream
    if len(tunes) > 1:   # sweep: best of three uncounted frames per setting
        for t in tunes:
            scene.set_tuning(*[int(x) for x in t.split(",")])
            best = min(scene.render_tiles_device(passes, d_accum.data_ptr(), d_rgba.data_ptr(), mode=mode,
                                                 stream=stream, want_stats=True)["kernel_ms"] for _ in range(3))
            print(f"tune {t:20s} kernel_ms {best:.2f}", flush=True)
        return
    for counted in (False, True):
        scene.set_counters(counted)
        st = scene.render_tiles_device(passes, d_accum.data_ptr(), d_rgba.data_ptr(), mode=mode,
                                       stream=stream, want_stats=True)
        r = max(st["rays"], 1)
        print(f"counted={counted} rays {st['rays']} kernel_ms {st['kernel_ms']:.2f} total_ms {st['total_ms']:.2f} "
              f"nodes/ray {st['nodes_visited'] / r:.2f} tris/ray {st['tris_fetched'] / r:.2f} "
              f"tested/ray {st['tris_tested'] / r:.2f}", flush=True)


if __name__ == "__main__":
    main()
