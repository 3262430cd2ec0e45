"""One process, one GPU, a `nccl` (= RCCL) process group of world size 1 with
its device given: drives the exchange helpers of raytracer_amd/tiles.py through
their COLLECTIVE branches on cuda tensors -- gather to rank 0, the all-gather of
tile ids plus the batched send / receive pair (to and from this very rank), the
counter all-reduce -- on a Cornell frame rendered by the real renderer, and
compares every composed frame with the plain paste of the same tile buffer.
Started by tests/test_gpu_tiles_dist.py.  Exit code 0 and a line "rccl ok" on
success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import gpu_dist_worker as wk
    from raytracer_amd import api, scenes, tiles
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        scene, _keep = wk.build_scene(api)
        tile_list = scenes.tiles(wk.W, wk.H, wk.GRID)
        stream = torch.cuda.current_stream().cuda_stream
        d_accum = torch.empty((wk.W * wk.H * 3,), dtype=torch.float32, device="cuda")
        d_tiles = torch.zeros((wk.W * wk.H * 4,), dtype=torch.uint8, device="cuda")
        st = scene.render_tiles_device(wk.make_passes(api, scenes, tile_list), d_accum.data_ptr(),
                                       d_tiles.data_ptr(), stream=stream, want_stats=True)
        want = torch.zeros((wk.H, wk.W, 4), dtype=torch.uint8, device="cuda")
        tiles.paste_tiles(want, d_tiles, tile_list)
        assert int(want[..., :3].max()) > 60, "the frame is black"

        # static deal: dist.gather on cuda tensors over RCCL
        got = torch.zeros_like(want)
        tiles.gather_static(d_tiles, tile_list, 1, 0, got, force_collective=True)
        torch.cuda.synchronize()
        assert torch.equal(got, want), "gather_static over RCCL differs from the plain paste"

        # pull queue: ids all-gathered, pixels sent and received in one batched RCCL group.
        # The tiles are "pulled" in a scrambled order, so the buffer is laid out in that order.
        order = [(i * 7) % len(tile_list) for i in range(len(tile_list))]
        scrambled = torch.zeros_like(d_tiles)
        offs, _total = tiles.tile_pixel_offsets(tile_list)
        at = 0
        for i in order:
            n = tile_list[i][3] * tile_list[i][4] * 4
            scrambled[at:at + n] = d_tiles[offs[i] * 4:offs[i] * 4 + n]
            at += n
        got = torch.zeros_like(want)
        tiles.gather_dynamic(scrambled, order, tile_list, 1, 0, got, force_collective=True)
        torch.cuda.synchronize()
        assert torch.equal(got, want), "gather_dynamic over RCCL differs from the plain paste"

        # counters: all-reduce on a cuda tensor
        total, flag = tiles.sum_counters([st["rays"], 1], "cuda", 1, force_collective=True)
        assert total == st["rays"] and flag == 1, (total, st["rays"], flag)
        print(f"rccl ok: {st['rays']} rays, frame {wk.W}x{wk.H}, backend {dist.get_backend()}", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
