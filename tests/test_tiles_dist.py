"""The N>1 path on CPU: two gloo ranks shard the frontend's tile grid,
"render" deterministic tile images, and bring them to rank 0 through the same
helpers bench.py uses on GPUs (static block-cyclic gather, dynamic pull queue +
reduce, counter all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import scenes, tiles

W, H, GRID = 250, 250, 4


def _expected_frame():
    yy, xx = np.mgrid[0:H, 0:W]
    f = np.zeros((H, W, 4), np.uint8)
    for tid, x, y, w, h in scenes.tiles(W, H, GRID):
        f[y:y + h, x:x + w, 0] = (xx[y:y + h, x:x + w] * 7 + tid) % 251
        f[y:y + h, x:x + w, 1] = (yy[y:y + h, x:x + w] * 3 + tid * 5) % 253
        f[y:y + h, x:x + w, 2] = tid
        f[y:y + h, x:x + w, 3] = 255
    return f


def _fake_render(tile):
    tid, x, y, w, h = tile
    return torch.from_numpy(_expected_frame()[y:y + h, x:x + w].copy())


def _worker(rank, world, port, dynamic, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tile_list = scenes.tiles(W, H, GRID)
        rays_here = 0
        if not dynamic:
            mine = [tile_list[i] for i in tiles.static_assignment(len(tile_list), world, rank)]
            px = [sum(t[3] * t[4] for t in tile_list[r::world]) for r in range(world)]
            buf = torch.zeros(max(px) * 4, dtype=torch.uint8)
            off = 0
            for t in mine:
                img = _fake_render(t).reshape(-1)
                buf[off:off + img.numel()] = img
                off += img.numel()
                rays_here += t[3] * t[4]
            frame = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
            frame = tiles.gather_static(buf, tile_list, world, rank, frame)
            counts = None
        else:
            store = dist.distributed_c10d._get_default_store()
            q = tiles.TileQueue(store, len(tile_list), chunk=3)
            q.reset(epoch=0)
            done = torch.zeros(len(tile_list), dtype=torch.int64)
            buf = torch.zeros(W * H * 4, dtype=torch.uint8)      # a rank may end up with any tile
            my_ids, off = [], 0
            while True:
                got = q.pull()
                if not got:
                    break
                for i in got:
                    tid, x, y, w, h = tile_list[i]
                    img = _fake_render(tile_list[i]).reshape(-1)
                    buf[off:off + img.numel()] = img
                    off += img.numel()
                    my_ids.append(i)
                    done[i] += 1
                    rays_here += w * h
            frame = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
            frame = tiles.gather_dynamic(buf, my_ids, tile_list, world, rank, frame)
            dist.all_reduce(done)
            counts = done
        total, = tiles.sum_counters([rays_here], "cpu", world)
        if rank == 0:
            np.save(out_path, frame.numpy())
            assert total == W * H, total
            if counts is not None:
                assert counts.tolist() == [1] * len(tile_list), counts.tolist()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("dynamic", [False, True])
def test_two_ranks_compose_the_frame(tmp_path, dynamic):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), dynamic, out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), _expected_frame())


def test_tile_grid_matches_the_frontend_rule():
    """Renderer.js:388-405: floor-sized tiles, remainder on the last row/column."""
    tl = scenes.tiles(250, 250, 4)
    assert len(tl) == 16
    assert [t[3] for t in tl[:4]] == [62, 62, 62, 64] and tl[15][3:] == (64, 64)
    assert tl[5][:3] == (5, 62, 62)
    cover = np.zeros((250, 250), int)
    for _, x, y, w, h in tl:
        cover[y:y + h, x:x + w] += 1
    assert (cover == 1).all()
    assert tiles.static_assignment(16, 4, 1) == [1, 5, 9, 13]
    offs, total = tiles.tile_pixel_offsets(tl[:3])
    assert offs == [0, 62 * 62, 2 * 62 * 62] and total == 3 * 62 * 62


def test_single_rank_paths():
    tile_list = scenes.tiles(W, H, GRID)
    buf = torch.cat([_fake_render(t).reshape(-1) for t in tile_list])
    frame = tiles.gather_static(buf, tile_list, 1, 0, torch.zeros((H, W, 4), dtype=torch.uint8))
    assert np.array_equal(frame.numpy(), _expected_frame())
    assert tiles.sum_counters([5, 7], "cpu", 1) == [5, 7]


def test_uniform_paste_equals_tile_loop():
    """The one-copy paste of a rank's tiles equals pasting them one by one, and
    declines tilings it does not cover (remainder tiles)."""
    import torch
    from raytracer_amd import scenes, tiles
    for (W, H, n, world) in [(64, 48, 4, 1), (64, 48, 4, 3), (60, 40, 4, 2), (50, 30, 4, 2)]:
        tl = scenes.tiles(W, H, n)
        for rank in range(world):
            mine = tl[rank::world]
            total = sum(t[3] * t[4] * 4 for t in mine)
            buf = torch.randint(0, 255, (total,), dtype=torch.uint8)
            want = torch.zeros((H, W, 4), dtype=torch.uint8)
            off = 0
            for (_t, x, y, w, h) in mine:
                want[y:y + h, x:x + w] = buf[off:off + w * h * 4].view(h, w, 4)
                off += w * h * 4
            got = torch.zeros((H, W, 4), dtype=torch.uint8)
            ok = tiles._paste_uniform(got, buf, mine)
            uniform = W % n == 0 and H % n == 0
            assert ok == uniform or (ok and all(t[3] == mine[0][3] and t[4] == mine[0][4] for t in mine))
            if ok:
                assert torch.equal(got, want)
