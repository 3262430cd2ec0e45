"""GPU parity of the batch ray cast (rt_trace_rays through the C ABI) against
the oracle: hit ids, t, u, v bit-exact in the float32 and float64 modes, the
traversal counters, and the edge cases of the domain."""
import os

import numpy as np
import pytest

from raytracer_amd import api, scenes
from util import (classify_fast_differences, cornell, oracle_scene, product_mesh, random_rays, rays6,
                  record_parity_counts)

pytestmark = pytest.mark.gpu


def _check(hits, ref, exact_uv=True):
    assert np.array_equal(hits["tri_pos"], ref["tri"])
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    if exact_uv:
        assert np.array_equal(hits["u"].view(np.uint32), ref["u"].view(np.uint32))
        assert np.array_equal(hits["v"].view(np.uint32), ref["v"].view(np.uint32))


@pytest.fixture(scope="module")
def soup20k(product_lib, oracle, gpu_device):
    sc = scenes.soup(20000, seed=0x5EED0001)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 32)
    return sc, mesh, bvh, api.Scene(mesh, bvh, gpu_device), oracle_scene(oracle, sc, bvh_export=bvh.export())


@pytest.mark.parametrize("mode", [api.MODE_STRICT_F32, api.MODE_FP64])
def test_incoherent_rays_bit_exact(soup20k, mode):
    _, _, _, scene, osc = soup20k
    rays = random_rays(200000, seed=5)
    rays["tmax"][::5] = np.float32(0.75)
    hits, st = scene.trace_rays(rays, mode=mode, counters=True)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax, mode=mode, counters=True)
    _check(hits, ref)
    c = ref["counters"]
    assert (st["rays"], st["nodes_visited"], st["box_tests"], st["tris_fetched"], st["tris_tested"]) == \
        (c["rays"], c["nodes_visited"], c["box_tests"], c["tris_fetched"], c["tris_tested"])
    assert (hits["tri_pos"] >= 0).mean() > 0.02


def test_primary_rays_config1_shape(soup20k):
    """Coherent camera rays (the 100k-soup / 1080p configuration at reduced
    size): float32 bit-exact, and the float64 validation mode differs from the
    float32 oracle only on a counted set of edge-grazing rays."""
    _, _, _, scene, osc = soup20k
    o, d = scenes.primary_rays(480, 270, {"x": 0.0, "y": 0.0, "z": 4.0})
    rays = api.make_rays(o, d)
    r6, tmax = rays6(rays)
    ref32 = osc.trace_rays(r6, tmax, mode=0)
    hits32, _ = scene.trace_rays(rays, mode=api.MODE_STRICT_F32)
    _check(hits32, ref32)
    hits64, _ = scene.trace_rays(rays, mode=api.MODE_FP64)
    ref64 = osc.trace_rays(r6, tmax, mode=1)
    _check(hits64, ref64)
    grazing = int((hits64["tri_pos"] != ref32["tri"]).sum())
    record_parity_counts("soup20k_480x270_primary_fp64_vs_f32_oracle", {"rays": int(len(rays)), "differing": grazing})
    assert grazing <= 0.001 * len(rays), grazing
    same = hits64["tri_pos"] == ref32["tri"]
    hit = same & (ref32["tri"] >= 0)
    rel = np.abs(hits64["t"][hit] - ref32["t"][hit]) / ref32["t"][hit]
    assert rel.max() < 1e-5


def test_empty_and_ragged_batches(soup20k):
    _, _, _, scene, osc = soup20k
    hits, st = scene.trace_rays(np.zeros(0, api.RAY_DTYPE))
    assert len(hits) == 0 and st["rays"] == 0
    for n in (1, 63, 64, 65, 257, 1000):
        rays = random_rays(n, seed=n)
        hits, _ = scene.trace_rays(rays)
        r6, tmax = rays6(rays)
        _check(hits, osc.trace_rays(r6, tmax))


def test_miss_reports_tmax_and_minus_one(soup20k):
    _, _, _, scene, _ = soup20k
    rays = api.make_rays([[50, 50, 50]] * 3, [[1, 0, 0]] * 3, tmax=[api.F32_MAX, 2.5, 0.0])
    hits, _ = scene.trace_rays(rays)
    assert hits["tri_pos"].tolist() == [-1, -1, -1]
    assert hits["t"].tolist() == [float(api.F32_MAX), 2.5, 0.0]


def test_single_leaf_root_and_tiny_scenes(product_lib, oracle, gpu_device):
    for n, use_bvh in [(1, True), (5, True), (40, False)]:
        sc = scenes.soup(n, seed=31 + n, extent=0.3, size=0.4)
        mesh = product_mesh(sc)
        bvh = api.Bvh.build(mesh, use_bvh, 6, 16)
        scene = api.Scene(mesh, bvh, gpu_device)
        osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
        rays = random_rays(5000, seed=n, extent=1.0)
        hits, _ = scene.trace_rays(rays)
        r6, tmax = rays6(rays)
        ref = osc.trace_rays(r6, tmax)
        _check(hits, ref)
        assert (hits["tri_pos"] >= 0).sum() > 0


def test_axis_aligned_rays_on_cornell(product_lib, oracle, gpu_device):
    """+-0 direction components (Inf inverse, NaN slabs) and exact box-face
    ties on the axis-aligned Cornell walls."""
    sc = cornell()
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 2, 16)
    scene = api.Scene(mesh, bvh, gpu_device)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    o, d = [], []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            for k in range(200):
                o.append([-2.5 + 0.025 * k, 0.1 + 0.025 * k, -5.0 + 0.02 * k])
                dr = np.zeros(3, np.float32)
                dr[ax] = sgn
                d.append(dr * np.float32(-1) if k % 2 else dr)
    rays = api.make_rays(np.array(o, np.float32), np.array(d, np.float32))
    hits, _ = scene.trace_rays(rays)
    r6, tmax = rays6(rays)
    _check(hits, osc.trace_rays(r6, tmax))
    assert (hits["tri_pos"] >= 0).sum() > 300


def test_subnormal_and_huge_values(product_lib, oracle, gpu_device):
    """Denormal direction components must not be flushed; huge coordinates must
    overflow the same way."""
    sc = scenes.soup(2000, seed=12)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 24)
    scene = api.Scene(mesh, bvh, gpu_device)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    rays = random_rays(4000, seed=3)
    rays["dir"][::4, 0] = np.float32(1e-41)
    rays["dir"][1::4, 1] = np.float32(-1e-42)
    rays["origin"][2::16] *= np.float32(1e30)
    rays["dir"][3::16] *= np.float32(1e20)
    hits, _ = scene.trace_rays(rays)
    r6, tmax = rays6(rays)
    _check(hits, osc.trace_rays(r6, tmax))


def test_device_pointer_entry_point(soup20k):
    import torch
    _, _, _, scene, osc = soup20k
    rays = random_rays(100000, seed=77)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
    d_hits = torch.zeros((len(rays), 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scene.trace_rays_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
    r6, tmax = rays6(rays)
    _check(hits, osc.trace_rays(r6, tmax))


def test_scene_info_and_errors(soup20k, gpu_device):
    sc, mesh, bvh, scene, _ = soup20k
    info = scene.info()
    assert info["n_tris"] == 20000 and info["n_nodes"] == bvh.n_nodes
    assert info["node_bytes"] == 32 * bvh.n_nodes and info["tri_bytes"] == 48 * 20000
    with pytest.raises(api._capi.RtError):
        api.Scene(mesh, bvh, 999)
    with pytest.raises(api._capi.RtError):
        scene.trace_rays(random_rays(10, seed=1), mode=17)


# ---------------------------------------------------------------------------
# fast mode (ordered traversal): deterministic, equal to its lane-serial CPU
# form bit for bit; equal to the reference order except for a counted handful
# of rays where the reference's own box pruning rejects a triangle it accepts
# ---------------------------------------------------------------------------
# the two forms of the ordered step: (mode, near-origin form of the lane-serial code)
FAST_FORMS = [(api.MODE_FAST_F32, False), (api.MODE_FAST_NEAR_F32, True)]


def _emul_fast(emul, bvh, rays, near=False):
    import ctypes as C
    rec, root = bvh.flatten_ordered()
    _, tris, _ = bvh.flatten()
    hits = np.zeros(rays.shape[0], api.HIT_DTYPE)
    cnt = np.zeros(5, np.uint64)
    emul.emul_trace_fast(rec.ctypes.data_as(C.c_void_p), C.c_uint32(root),
                         tris.ctypes.data_as(C.c_void_p), C.c_int32(tris.shape[0]),
                         rays.ctypes.data_as(C.c_void_p), C.c_int64(rays.shape[0]),
                         hits.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), C.c_int(int(near)))
    return hits, cnt


@pytest.mark.parametrize("mode,near", FAST_FORMS)
def test_fast_mode_matches_cpu_form_and_reference(soup20k, emul, oracle, mode, near):
    sc, _, bvh, scene, osc = soup20k
    rays = random_rays(300000, seed=9)          # origins within 1.5 of the centre: inside the near region
    rays["tmax"][::7] = np.float32(0.9)
    hits, st = scene.trace_rays(rays, mode=mode, counters=True)
    cpu, cnt = _emul_fast(emul, bvh, rays, near)
    assert np.array_equal(hits.view(np.uint32), cpu.view(np.uint32))
    assert [st["rays"], st["nodes_visited"], st["box_tests"], st["tris_fetched"], st["tris_tested"]] == cnt.tolist()
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax, counters=True)
    counts = classify_fast_differences(oracle, sc["verts"], bvh.export()["order"], rays, hits, ref["tri"], ref["t"])
    record_parity_counts("soup20k_300k_incoherent_rays_fast_vs_oracle" + ("_near_form" if near else ""), counts)
    assert counts["differing"] <= 3, counts
    same = hits["tri_pos"] == ref["tri"]
    assert np.array_equal(hits["t"][same].view(np.uint32), ref["t"][same].view(np.uint32))
    c = ref["counters"]
    assert 56 * st["nodes_visited"] + 48 * st["tris_fetched"] < 56 * c["nodes_visited"] + 48 * c["tris_fetched"]


def test_fast_mode_with_point_triangles(product_lib, emul, gpu_device):
    """Triangles collapsed to points (some identical) give records without
    extent; unused slots of such records carry an inverted box that rounding can
    collapse for far-away origins.  The kernel and its lane-serial CPU form
    must still agree bit for bit, counters included."""
    sc = scenes.soup(600, seed=31)
    v = sc["verts"].reshape(-1, 3, 3).copy()
    v[100:140] = v[100:140, :1, :]
    v[140:150] = v[140, 0][None, None, :]
    v[150:153] = np.float32(0.0)
    sc = dict(sc, verts=np.ascontiguousarray(v.reshape(-1, 9)))
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 2, 24)
    scene = api.Scene(mesh, bvh, gpu_device)
    rays = random_rays(50000, seed=9)
    rays["origin"][::3] *= np.float32(1e6)
    hits, st = scene.trace_rays(rays, mode=api.MODE_FAST_F32, counters=True)
    cpu, cnt = _emul_fast(emul, bvh, rays)
    assert np.array_equal(hits.view(np.uint32), cpu.view(np.uint32))
    assert [st["rays"], st["nodes_visited"], st["box_tests"], st["tris_fetched"], st["tris_tested"]] == cnt.tolist()
    strict, _ = scene.trace_rays(rays, mode=api.MODE_STRICT_F32)
    assert np.array_equal(hits["tri_pos"], strict["tri_pos"])


@pytest.mark.parametrize("mode,near", FAST_FORMS)
def test_fast_mode_on_axis_aligned_geometry(product_lib, oracle, gpu_device, mode, near):
    """Rays along quad diagonals and box faces: exact t ties between the two
    triangles of a wall resolve to the earlier array position, as in the
    reference order."""
    sc = scenes.room_mesh(48, seed=5, total=6000)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 4, 40)
    scene = api.Scene(mesh, bvh, gpu_device)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    cam = sc["camera"]
    o, d = scenes.primary_rays(320, 180, cam)
    extra_o, extra_d = [], []
    for k in range(2000):           # rays aimed exactly at wall diagonals
        s = k / 2000.0
        extra_o.append([0.0, 1.0, 0.0])
        tgt = np.array([-1.38 + 2.76 * s, 1.08 * s, -1.38], np.float32)
        v = tgt - np.float32([0.0, 1.0, 0.0])
        extra_d.append(v / np.linalg.norm(v))
    rays = np.concatenate([api.make_rays(o, d),
                           api.make_rays(np.array(extra_o, np.float32), np.array(extra_d, np.float32)),
                           random_rays(100000, seed=2, extent=1.0)])
    hits, _ = scene.trace_rays(rays, mode=mode)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax)
    counts = classify_fast_differences(oracle, sc["verts"], bvh.export()["order"], rays, hits, ref["tri"], ref["t"])
    record_parity_counts("room6k_axis_aligned_fast_vs_oracle" + ("_near_form" if near else ""), counts)
    assert counts["differing"] <= 1e-4 * len(rays), counts
    same = (hits["tri_pos"] == ref["tri"])
    assert np.array_equal(hits["t"][same].view(np.uint32), ref["t"][same].view(np.uint32))
    assert (hits["tri_pos"] >= 0).mean() > 0.5


def test_fast_mode_deep_and_degenerate_trees(product_lib, oracle, gpu_device):
    for n, use_bvh, leaf, depth in [(1, True, 6, 16), (300, False, 6, 16), (4000, True, 1, 40),
                                    (4000, True, 6, 2)]:
        sc = scenes.soup(n, seed=50 + n + depth, extent=0.5, size=0.1)
        mesh = product_mesh(sc)
        bvh = api.Bvh.build(mesh, use_bvh, leaf, depth)
        scene = api.Scene(mesh, bvh, gpu_device)
        osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
        rays = random_rays(20000, seed=n, extent=1.0)
        r6, tmax = rays6(rays)
        ref = osc.trace_rays(r6, tmax)
        for mode, _near in FAST_FORMS:
            hits, _ = scene.trace_rays(rays, mode=mode)
            counts = classify_fast_differences(oracle, sc["verts"], bvh.export()["order"], rays, hits, ref["tri"], ref["t"])
            assert counts["differing"] <= 2, counts
            assert (hits["tri_pos"] >= 0).sum() > 0


@pytest.mark.parametrize("scale,offset", [(1.0e6, 0.0), (0.6, 0.0), (1.0, 4.0e4), (1.0e3, -7.0e6)])
def test_fast_mode_at_extreme_scales(product_lib, oracle, gpu_device, emul, scale, offset):
    """The 8-bit child boxes of the ordered view are relative to a per-record
    frame: scenes that are huge, tiny or far from the origin (where one step of
    the frame is many float32 ulps, or less than one) must still give the CPU
    form's hits bit for bit and the reference order's hits up to a counted
    handful; rays with denormal direction components included."""
    sc = scenes.soup(6000, seed=77, extent=1.0, size=0.08)
    sc["verts"] = (sc["verts"] * np.float32(scale) + np.float32(offset)).astype(np.float32)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 32)
    scene = api.Scene(mesh, bvh, gpu_device)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    rays = random_rays(60000, seed=4, extent=1.2)
    rays["origin"] = (rays["origin"] * np.float32(scale) + np.float32(offset)).astype(np.float32)
    rays["dir"][::9, 0] = np.float32(1e-41)
    rays["dir"][4::9, 2] = np.float32(0.0)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax)
    for mode, near in FAST_FORMS:          # the rays start inside the near region whatever the scale
        hits, _ = scene.trace_rays(rays, mode=mode)
        cpu, _ = _emul_fast(emul, bvh, rays, near)
        assert np.array_equal(hits.view(np.uint32), cpu.view(np.uint32))
        counts = classify_fast_differences(oracle, sc["verts"], bvh.export()["order"], rays, hits, ref["tri"],
                                           ref["t"], degenerate_ok=True)
        record_parity_counts(f"soup6k_scale_{scale:g}_offset_{offset:g}_fast_vs_oracle" + ("_near_form" if near else ""),
                             counts)
        differ = hits["tri_pos"] != ref["tri"]
        degenerate = np.zeros(len(rays), bool)
        degenerate[::9] = True
        degenerate[4::9] = True
        assert int((differ & ~degenerate).sum()) <= 3
        # A zero or denormal direction component gives an infinite inverse direction;
        # where such a ray starts exactly on a box plane (common when the coordinates
        # are coarse in float32) the product is NaN, which the reference's
        # compare-and-assign slab (aabb.go:71-96) and the min/max slab of the ordered
        # traversal treat differently: a small counted share of those rays.
        assert int((differ & degenerate).sum()) <= max(3, int(0.002 * degenerate.sum()))
        same = ~differ
        assert np.array_equal(hits["t"][same].view(np.uint32), ref["t"][same].view(np.uint32))
    # the scale must not make the scene invisible to the test
    assert (ref["tri"] >= 0).sum() > 100


@pytest.mark.parametrize("n,use_bvh,leaf,depth", [(300, False, 6, 16), (4000, True, 200, 4), (4000, True, 1, 40),
                                                  (2, True, 1, 4), (5000, True, 127, 3), (5000, True, 128, 3)])
def test_reference_order_views(product_lib, oracle, gpu_device, n, use_bvh, leaf, depth):
    """The reference-order kernel reads a linked copy of the node array whose
    top levels sit in LDS; trees with a leaf of more than 127 triangles do not
    fit that copy and are read from the plain preorder array.  Both must give
    the oracle's hits and counters bit for bit: one big leaf, depth-capped
    trees with large leaves on either side of the limit, single-triangle
    leaves, a two-triangle scene."""
    sc = scenes.soup(n, seed=900 + n + leaf, extent=0.6, size=0.1)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, use_bvh, leaf, depth)
    scene = api.Scene(mesh, bvh, gpu_device)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    rays = random_rays(20000, seed=n + depth, extent=1.0)
    rays["tmax"][::5] = np.float32(0.8)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax, counters=True)
    for mode in (api.MODE_STRICT_F32, api.MODE_FP64):
        hits, st = scene.trace_rays(rays, mode=mode, counters=True)
        if mode == api.MODE_STRICT_F32:
            _check(hits, ref)
            c = ref["counters"]
            assert (st["nodes_visited"], st["box_tests"], st["tris_fetched"], st["tris_tested"]) == \
                (c["nodes_visited"], c["box_tests"], c["tris_fetched"], c["tris_tested"])
        else:
            assert (hits["tri_pos"] != ref["tri"]).mean() < 0.01
    assert (ref["tri"] >= 0).sum() > 0


def test_tuning_does_not_change_results(soup20k):
    """Stack size, refill threshold, grid size and the two phase thresholds only
    steer scheduling: hits and counters stay identical in both kernels."""
    _, _, _bvh, scene, _osc = soup20k
    rays = random_rays(120000, seed=21)
    rays["tmax"][::3] = np.float32(1.1)
    try:
        base = {}
        for mode in (api.MODE_STRICT_F32, api.MODE_FAST_F32):
            scene.set_tuning(8, 0, 8, 0, 0)
            base[mode] = scene.trace_rays(rays, mode=mode, counters=True)
        for tune in [(16, 64, 1, 64, 64), (24, 1, 3, 1, 1), (12, 33, 16, 17, 5), (8, 60, 8, 32, 16)]:
            scene.set_tuning(*tune)
            for mode in (api.MODE_STRICT_F32, api.MODE_FAST_F32):
                hits, st = scene.trace_rays(rays, mode=mode, counters=True)
                assert np.array_equal(hits.view(np.uint32), base[mode][0].view(np.uint32)), (tune, mode)
                for key in ("nodes_visited", "box_tests", "tris_fetched", "tris_tested"):
                    assert st[key] == base[mode][1][key], (tune, mode, key)
    finally:
        scene.set_tuning(8, 0, 8, 0, 0)


def test_trace_calls_from_two_threads_on_two_streams(soup20k):
    """rt_trace_rays_device is asynchronous and keeps no per-call state in the
    scene beyond a ray cursor of its own: two host threads launching on two
    streams at once (many short launches, both modes) get the hits of a call
    made alone."""
    import threading
    import torch
    _, _, _, scene, _ = soup20k
    sets = [random_rays(200_000, seed=31), random_rays(150_000, seed=32, extent=1.0)]
    modes = [api.MODE_STRICT_F32, api.MODE_FAST_F32]
    alone = [scene.trace_rays(r, mode=m)[0] for r, m in zip(sets, modes)]
    d_rays = [torch.from_numpy(r.view(np.float32).reshape(-1, 8).copy()).cuda() for r in sets]
    d_hits = [[torch.zeros((len(r), 4), dtype=torch.float32, device="cuda") for _ in range(40)] for r in sets]
    streams = [torch.cuda.Stream() for _ in sets]
    errors = []

    def work(i):
        try:
            for k in range(40):
                scene.trace_rays_device(d_rays[i].data_ptr(), len(sets[i]), d_hits[i][k].data_ptr(),
                                        mode=modes[i], stream=streams[i].cuda_stream)
        except Exception as e:   # noqa: BLE001 - reported below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(2):
        want = alone[i].view(np.uint32).reshape(-1, 4)
        for k in (0, 17, 39):
            assert np.array_equal(d_hits[i][k].cpu().numpy().view(np.uint32), want), (i, k)


def test_trace_call_captured_into_a_graph_replays(soup20k):
    """rt_trace_rays_device allocates nothing and never synchronises, so it can be
    captured into a HIP graph; every replay must trace the rays again (the ray
    cursor is reset by a kernel of the call, not by a memset node)."""
    import torch
    _, _, _, scene, _ = soup20k
    rays = random_rays(150_000, seed=41)
    want = {m: scene.trace_rays(rays, mode=m)[0] for m in (api.MODE_STRICT_F32, api.MODE_FAST_F32)}
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
    for mode, ref in want.items():
        d_hits = torch.zeros((len(rays), 4), dtype=torch.float32, device="cuda")
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            scene.trace_rays_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), mode=mode,
                                    stream=torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            d_hits.zero_()
            graph.replay()
            torch.cuda.synchronize()
            got = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mode


def test_short_reciprocal_of_the_triangle_test_is_the_ieee_quotient(product_lib, gpu_device):
    """rt_core.h: recip_det (1 / det of the triangle test, triangle.go:86) and recip3
    (a ray's inverse direction, ray.go:33-37) take the hardware reciprocal plus two
    fused Newton steps for operands of magnitude [2^-100, 2^100] instead of the
    ten-instruction division.  tools/ubench/rcp_check compares that sequence with
    1.0f / x for every such float32 of either sign on the GPU it runs on: no
    mismatch allowed (the bit-exact modes need the IEEE quotient)."""
    import json
    import subprocess
    from raytracer_amd import build
    exe = build.RCP_CHECK_BIN
    assert os.path.exists(exe), "tools/ubench/rcp_check is built by __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr)
    doc = json.loads(out.stdout.strip().splitlines()[-1])
    assert doc["mismatches"] == 0 and doc["floats_checked"] > 3_300_000_000
