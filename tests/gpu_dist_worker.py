"""One rank of the two-rank tile test (started by tests/test_gpu_tiles_dist.py,
one process per rank, gloo, every rank on GPU 0): renders this rank's share of
a 4x4-tile Cornell frame with the real renderer and takes part in the exchange
that brings the tiles to rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W = H = 250
GRID = 4
CAM = {"x": -0.225, "y": 2.55, "z": 6.0}


def make_passes(api, scenes, group):
    return [api.pass_from_dict(scenes.make_pass_dict(
        W, H, CAM, spp=3, bounces=3, light_samples=2, intensity=100.0, seed=1 + t[0],
        x_offset=t[1], y_offset=t[2], tile_w=t[3], tile_h=t[4], task_id=t[0])) for t in group]


def build_scene(api):
    from util import cornell, product_mesh
    sc = cornell()
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 16)
    return api.Scene(mesh, bvh, 0), (mesh, bvh)


def main():
    out_path, dynamic = sys.argv[1], sys.argv[2] == "1"
    import torch
    import torch.distributed as dist
    from raytracer_amd import api, scenes, tiles
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        scene, _keep = build_scene(api)
        tile_list = scenes.tiles(W, H, GRID)
        stream = torch.cuda.current_stream().cuda_stream
        d_accum = torch.empty((W * H * 3,), dtype=torch.float32, device="cuda")
        d_tiles = torch.zeros((W * H * 4,), dtype=torch.uint8, device="cuda")
        frame = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
        rays = 0
        if not dynamic:
            mine = [tile_list[i] for i in tiles.static_assignment(len(tile_list), world, rank)]
            st = scene.render_tiles_device(make_passes(api, scenes, mine), d_accum.data_ptr(), d_tiles.data_ptr(),
                                           stream=stream, want_stats=True)
            rays = st["rays"]
            px = [sum(t[3] * t[4] for t in tile_list[r::world]) for r in range(world)]
            torch.cuda.synchronize()
            frame = tiles.gather_static(d_tiles[:max(px) * 4].cpu(), tile_list, world, rank, frame)
        else:
            q = tiles.TileQueue(dist.distributed_c10d._get_default_store(), len(tile_list), chunk=3)
            q.reset(epoch=0)
            my_ids, off = [], 0
            while True:
                got = q.pull()
                if not got:
                    break
                group = [tile_list[i] for i in got]
                st = scene.render_tiles_device(make_passes(api, scenes, group), d_accum.data_ptr() + off * 12,
                                               d_tiles.data_ptr() + off * 4, stream=stream, want_stats=True)
                rays += st["rays"]
                my_ids += got
                off += sum(t[3] * t[4] for t in group)
            torch.cuda.synchronize()
            frame = tiles.gather_dynamic(d_tiles.cpu(), my_ids, tile_list, world, rank, frame)
        total, mine_max = tiles.sum_counters([rays, 1 if rays > 0 else 0], "cpu", world)
        if rank == 0:
            np.savez(out_path, frame=frame.numpy(), rays=np.int64(total), ranks_with_work=np.int64(mine_max))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
