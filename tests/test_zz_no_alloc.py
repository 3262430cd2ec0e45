"""rt_render_tile_static -- the entry point that allocates nothing (its path
workspace is the library's static device array) -- against
rt_render_tile_device, through the C ABI: same per-pixel sums and RGBA bytes,
bit for bit, on two calls in a row, the second with a smaller tile (the array
keeps the first call's contents), in the reference-order and the ordered mode.
Collected last on purpose: it runs after every other GPU test and must also
pass alone in a fresh process."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import api, scenes
from util import cornell, product_mesh

pytestmark = pytest.mark.gpu

CORNELL_CAM = {"x": -0.225, "y": 2.55, "z": 6.0}


def _pass(x, y, w, h, spp, L, B, seed):
    return api.pass_from_dict(scenes.make_pass_dict(
        250, 250, CORNELL_CAM, spp=spp, bounces=B, light_samples=L, intensity=100.0, seed=seed,
        x_offset=x, y_offset=y, tile_w=w, tile_h=h, task_id=3))


def _both(scene, p, mode):
    import torch
    n = p.width * p.height
    stream = torch.cuda.current_stream().cuda_stream
    acc_a = torch.full((n * 3,), -1.0, dtype=torch.float32, device="cuda")
    acc_b = torch.full((n * 3,), -2.0, dtype=torch.float32, device="cuda")
    img_a = torch.zeros((n * 4,), dtype=torch.uint8, device="cuda")
    img_b = torch.ones((n * 4,), dtype=torch.uint8, device="cuda")
    scene.render_tile_device(p, acc_a.data_ptr(), img_a.data_ptr(), mode=mode, stream=stream)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    scene.render_tile_static(p, acc_b.data_ptr(), img_b.data_ptr(), mode=mode, stream=stream)
    torch.cuda.synchronize()
    # (the allocating call took ~100 MB for this tile; 2 MB allows for the runtime's own bookkeeping)
    assert torch.cuda.mem_get_info()[0] >= free_before - (2 << 20), "the static call took device memory"
    assert torch.equal(acc_a.view(torch.int32), acc_b.view(torch.int32))
    assert torch.equal(img_a, img_b)
    return img_b.cpu().numpy()


@pytest.mark.parametrize("mode", [api.MODE_STRICT_F32, api.MODE_FAST_F32])
def test_static_workspace_call_equals_the_allocating_one(product_lib, gpu_device, mode):
    sc = cornell()
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 16)
    scene = api.Scene(mesh, bvh, gpu_device)
    # the reference's own tile shape: 62x62 pixels, 25 spp, 6 light samples, 3 bounces; its
    # 96 100 camera samples do not fit the array at once, so the samples go in several batches
    big = _pass(62, 62, 62, 62, 25, 6, 3, seed=1)
    assert 62 * 62 * 25 * api.workspace_bytes_per_path(6, 3) > api.STATIC_WORKSPACE_BYTES
    img = _both(scene, big, mode)
    assert img.reshape(62, 62, 4)[..., :3].max() > 60
    # second call, smaller input, other settings: nothing of the first call may leak in
    small = _pass(100, 130, 23, 17, 3, 2, 5, seed=7)
    _both(scene, small, mode)
    # and the first again, after the small one
    _both(scene, big, mode)


def test_static_workspace_limit_is_an_error(product_lib, gpu_device):
    import torch
    sc = cornell()
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 16)
    scene = api.Scene(mesh, bvh, gpu_device)
    per_path = api.workspace_bytes_per_path(1, 4)
    side = int((api.STATIC_WORKSPACE_BYTES / per_path) ** 0.5) + 8      # a few pixels too many
    p = api.pass_from_dict(scenes.make_pass_dict(side, side, CORNELL_CAM, spp=1, bounces=4,
                                                 light_samples=1, intensity=100.0, seed=1))
    acc = torch.zeros((side * side * 3,), dtype=torch.float32, device="cuda")
    img = torch.zeros((side * side * 4,), dtype=torch.uint8, device="cuda")
    with pytest.raises(api._capi.RtError) as e:
        scene.render_tile_static(p, acc.data_ptr(), img.data_ptr())
    assert e.value.status == -1 and "static workspace" in e.value.message
    # just below the limit it renders, and equals the allocating call
    side2 = int((api.STATIC_WORKSPACE_BYTES / per_path) ** 0.5) - 8
    p2 = api.pass_from_dict(scenes.make_pass_dict(side2, side2, CORNELL_CAM, spp=2, bounces=4,
                                                  light_samples=1, intensity=100.0, seed=1))
    _both(scene, p2, api.MODE_STRICT_F32)


def test_static_workspace_call_under_graph_capture(product_lib, gpu_device):
    """What the entry point is for: the whole tile render is captured into a HIP
    graph (a capture forbids allocation and host synchronisation) and replayed;
    every replay gives the bytes of the allocating call."""
    import torch
    sc = cornell()
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 16)
    scene = api.Scene(mesh, bvh, gpu_device)
    p = _pass(62, 124, 62, 62, 4, 6, 3, seed=3)
    n = p.width * p.height
    acc_ref = torch.zeros((n * 3,), dtype=torch.float32, device="cuda")
    img_ref = torch.zeros((n * 4,), dtype=torch.uint8, device="cuda")
    scene.render_tile_device(p, acc_ref.data_ptr(), img_ref.data_ptr(),
                             stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    acc = torch.zeros((n * 3,), dtype=torch.float32, device="cuda")
    img = torch.zeros((n * 4,), dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        scene.render_tile_static(p, acc.data_ptr(), img.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        img.zero_()
        acc.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(img, img_ref)
        assert torch.equal(acc.view(torch.int32), acc_ref.view(torch.int32))
    # an ordinary call after the capture still works (the event chain was left alone)
    _both(scene, p, api.MODE_STRICT_F32)
