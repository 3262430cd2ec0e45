"""Two ranks, one GPU: the real renderer on every rank's share of a 4x4-tile
Cornell frame, the tiles brought to rank 0 by the same exchange the benchmark
uses (gloo here, RCCL there).  The composed frame equals the frame one rank
renders alone byte for byte, the summed ray counters equal its ray count
(the frontend sums the workers' counters, Renderer.js:772-778), for the static
block-cyclic deal and for the pull queue.  Also: bench.py started plainly with
--gpus 2 launches two ranks itself and says so in its line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single_rank_frame():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_dist_worker as wk
    from raytracer_amd import api, scenes
    scene, _keep = wk.build_scene(api)
    tile_list = scenes.tiles(wk.W, wk.H, wk.GRID)
    d_accum = torch.empty((wk.W * wk.H * 3,), dtype=torch.float32, device="cuda")
    d_tiles = torch.zeros((wk.W * wk.H * 4,), dtype=torch.uint8, device="cuda")
    st = scene.render_tiles_device(wk.make_passes(api, scenes, tile_list), d_accum.data_ptr(),
                                   d_tiles.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                                   want_stats=True)
    torch.cuda.synchronize()
    from raytracer_amd import tiles
    frame = torch.zeros((wk.H, wk.W, 4), dtype=torch.uint8)
    tiles.paste_tiles(frame, d_tiles.cpu(), tile_list)
    return frame.numpy(), st["rays"]


@pytest.mark.parametrize("dynamic", [0, 1])
def test_two_ranks_on_one_gpu_compose_the_single_rank_frame(product_lib, gpu_device, tmp_path, dynamic):
    want, want_rays = _single_rank_frame()
    assert want[..., :3].max() > 60
    out = str(tmp_path / "frame.npz")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_dist_worker.py"),
                                       out, str(dynamic)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=240)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], logs
    got = np.load(out)
    assert np.array_equal(got["frame"], want)
    assert int(got["rays"]) == want_rays
    if not dynamic:
        assert int(got["ranks_with_work"]) == 2


def test_exchange_helpers_over_rccl_in_a_group_of_one(product_lib, gpu_device):
    """The RCCL branch of the exchange -- dist.gather, all_gather + one batched
    isend/irecv group, all_reduce, all on cuda tensors -- executed for real: a
    child process opens a `nccl` group of world size 1 on this GPU and forces
    the helpers through their collective paths (tests/gpu_rccl_worker.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_rccl_worker.py")], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "rccl ok" in out.stdout and "backend nccl" in out.stdout


def test_bench_starts_its_own_ranks(product_lib, gpu_device):
    """`python bench.py --gpus 2` with no launcher around it must run two ranks
    (never one rank labelled as such): here on one GPU over gloo, on the small
    functional workload."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for extra in ([], ["--dynamic"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                              "--workload", "small", "--steps", "1", "--warmup", "1", "--spp", "2",
                              "--cpu-seconds", "0"] + extra, env=env, capture_output=True, text=True, timeout=280)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        doc = json.loads(line)
        assert doc["n_gpus"] == 2 and doc["value"] > 0
        ranks = doc["ranks"]       # what each rank spent where, for attributing a shortfall at N ranks
        assert len(ranks["render_ms_per_step"]) == 2 and len(ranks["gather_ms_per_step"]) == 2
        assert sum(ranks["tiles_per_step"]) == doc["config"]["tiles"]
        assert abs(sum(ranks["rays_per_step"]) - doc["config"]["rays_per_step"]) < 1
        if extra:
            assert min(ranks["pulls_per_step"]) >= 1
        assert doc["config"]["tile_assignment"] == ("dynamic pull queue" if extra else "block-cyclic")
    # asking for more GPUs than there are over RCCL is refused, not downgraded
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "n_gpus" not in out.stdout
