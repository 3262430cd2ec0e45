#!/usr/bin/env python3
"""Experiment: does running two independent halves of the frame on two HIP
streams (two scene handles, two host threads) beat one stream?  The traversal
launches are VALU-bound, the shading launches HBM-bound; if the hardware
overlaps them the two-stream total is shorter."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from raytracer_amd import api, scenes
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bpc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    sc = scenes.room_mesh(706, seed=0x5EED0002, total=1_000_000)
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    bvh = api.Bvh.build(mesh, True, 8, 40)
    W, H = 3840, 2160
    tl = scenes.tiles(W, H, 16)

    def passes(tiles_):
        return api.pass_array([api.pass_from_dict(scenes.make_pass_dict(
            W, H, sc["camera"], spp=spp, bounces=4, light_samples=1, intensity=6.0, seed=1,
            x_offset=x, y_offset=y, tile_w=w, tile_h=h, task_id=tid)) for (tid, x, y, w, h) in tiles_])

    def run(n_streams):
        parts = [tl[i::n_streams] for i in range(n_streams)]
        scs, bufs, streams, arrs = [], [], [], []
        for p in parts:
            s = api.Scene(mesh, bvh, 0)
            s.set_tuning(8, 60, bpc, 32, 16)
            s.set_max_paths(1 << 25)
            px = sum(t[3] * t[4] for t in p)
            scs.append(s)
            bufs.append((torch.empty((px * 3,), dtype=torch.float32, device="cuda"),
                         torch.zeros((px * 4,), dtype=torch.uint8, device="cuda")))
            streams.append(torch.cuda.Stream())
            arrs.append(passes(p))
        out = [None] * n_streams

        def work(i):
            out[i] = scs[i].render_tiles_device(arrs[i], bufs[i][0].data_ptr(), bufs[i][1].data_ptr(),
                                                mode=api.MODE_FAST_F32, stream=streams[i].cuda_stream,
                                                want_stats=True)
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(i,)) for i in range(n_streams)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rays = sum(o["rays"] for o in out)
            if rep:
                best = max(best or 0, rays / dt / 1e6)
        return best

    for n in (1, 2, 3):
        print(f"{n} stream(s), {bpc} blocks/CU: {run(n):.1f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
