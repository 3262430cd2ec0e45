"""GPU checks at the full sizes of the benchmark configurations.

Config 1 (100 000-triangle soup, 1920x1080 primary rays) is small enough for
the oracle to trace every ray, so it is compared bit for bit.  The 1M-triangle
room at 3840x2160 is not: there the oracle renders a few whole tiles of the
frame (bit-exact comparison of the reference-order mode), and the rest is
covered by properties that do not depend on size -- the ordered mode agrees
with the reference order up to a counted handful of pixels, the frame does not
depend on how its tiles are grouped into calls, two runs give the same bytes,
every hit lies inside its triangle, and the per-tile ray counters add up.
"""
import numpy as np
import pytest

from raytracer_amd import api, scenes
from util import classify_fast_differences, oracle_scene, product_mesh, rays6, record_parity_counts

pytestmark = pytest.mark.gpu


def _tables():
    return api.default_sample_table(12345, 0xC0FFEE01), api.default_sample_table(100001, 0xC0FFEE02)


def test_config1_every_primary_ray_bit_exact(product_lib, oracle, gpu_device):
    """100 000 triangles (centroids U([-1,1]^3), offsets U([-0.02,0.02]^3), seed
    0x5EED0001), SAH leaf 6 depth 32, 1920x1080 rays through pixel centres from
    (0,0,4) looking down -Z, fov 45, tmax = MaxFloat32: 0 mismatching hit ids
    against the oracle, t/u/v bit-identical."""
    sc = scenes.soup(100_000, seed=0x5EED0001)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 32)
    scene = api.Scene(mesh, bvh, gpu_device)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    o, d = scenes.primary_rays(1920, 1080, {"x": 0.0, "y": 0.0, "z": 4.0})
    rays = api.make_rays(o, d)
    assert rays.shape[0] == 1920 * 1080
    hits, st = scene.trace_rays(rays, mode=api.MODE_STRICT_F32, counters=True)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax, counters=True)
    assert int((hits["tri_pos"] != ref["tri"]).sum()) == 0
    for f in ("t", "u", "v"):
        assert np.array_equal(hits[f].view(np.uint32), ref[f].view(np.uint32)), f
    c = ref["counters"]
    assert (st["nodes_visited"], st["tris_fetched"], st["tris_tested"]) == \
        (c["nodes_visited"], c["tris_fetched"], c["tris_tested"])
    hit_rate = float((hits["tri_pos"] >= 0).mean())
    assert 0.05 < hit_rate < 0.95
    # ordered mode on the same rays: a counted handful may differ, each of them a hit the
    # reference's own box test pruned; t equal where ids agree
    fast, _ = scene.trace_rays(rays, mode=api.MODE_FAST_F32)
    counts = classify_fast_differences(oracle, sc["verts"], bvh.export()["order"], rays, fast, ref["tri"], ref["t"])
    record_parity_counts("config1_fast_vs_oracle", counts)
    same = fast["tri_pos"] == hits["tri_pos"]
    assert counts["differing"] <= 20, counts
    assert np.array_equal(fast["t"][same].view(np.uint32), hits["t"][same].view(np.uint32))
    # the near-origin form of the ordered step (the camera lies within one scene diagonal of the soup)
    info = scene.info()
    assert all(info["near_lo"][a] <= rays["origin"][:, a].min() and rays["origin"][:, a].max() <= info["near_hi"][a]
               for a in range(3))
    near, _ = scene.trace_rays(rays, mode=api.MODE_FAST_NEAR_F32)
    counts = classify_fast_differences(oracle, sc["verts"], bvh.export()["order"], rays, near, ref["tri"], ref["t"])
    record_parity_counts("config1_fast_near_form_vs_oracle", counts)
    same = near["tri_pos"] == hits["tri_pos"]
    assert counts["differing"] <= 20, counts
    assert np.array_equal(near["t"][same].view(np.uint32), hits["t"][same].view(np.uint32))
    # float64 validation mode against the float32 reference: the counted edge-grazing set
    h64, _ = scene.trace_rays(rays, mode=api.MODE_FP64)
    r64 = osc.trace_rays(r6, tmax, mode=1)
    assert np.array_equal(h64["tri_pos"], r64["tri"])
    grazing = int((h64["tri_pos"] != ref["tri"]).sum())
    record_parity_counts("config1_fp64_vs_f32_oracle", {"rays": int(len(rays)), "differing": grazing})
    assert grazing <= 0.001 * len(rays), grazing


@pytest.fixture(scope="module")
def room_1m(product_lib, oracle, gpu_device):
    sc = scenes.room_mesh(706, seed=0x5EED0002, total=1_000_000)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 8, 40)
    scene = api.Scene(mesh, bvh, gpu_device)
    return sc, mesh, bvh, scene


W, H, GRID = 3840, 2160, 16


def _pass(sc, tile, spp=1, seed=1):
    tid, x, y, w, h = tile
    return api.pass_from_dict(scenes.make_pass_dict(
        W, H, sc["camera"], spp=spp, bounces=4, light_samples=1, intensity=6.0, seed=seed,
        x_offset=x, y_offset=y, tile_w=w, tile_h=h, task_id=tid))


def _pass_dict(sc, tile, spp=1, seed=1):
    tid, x, y, w, h = tile
    return scenes.make_pass_dict(W, H, sc["camera"], spp=spp, bounces=4, light_samples=1,
                                 intensity=6.0, seed=seed, x_offset=x, y_offset=y, tile_w=w,
                                 tile_h=h, task_id=tid)


def test_room_1m_tiles_bit_exact_against_oracle(room_1m, oracle):
    """Sixteen whole 240x135 tiles of the 4K frame (one per tile row, columns
    spread) over the 1M-triangle room, 4 bounces, 1 light sample: per-pixel
    sums, RGBA bytes and ray counts of the reference-order mode equal the
    oracle's."""
    sc, _mesh, bvh, scene = room_1m
    assert bvh.n_tris == 1_000_000
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    osc.set_sample_tables(*_tables())
    tl = scenes.tiles(W, H, GRID)
    for tile in [tl[r * GRID + (r * 5) % GRID] for r in range(GRID)]:
        out = scene.render_tile(_pass(sc, tile))
        ref = osc.render_tile(oracle.pass_from_dict(_pass_dict(sc, tile)))
        assert np.array_equal(out["accum"].view(np.uint32), ref["accum"].view(np.uint32))
        assert np.array_equal(out["rgba"], ref["rgba"])
        assert out["stats"]["rays"] == ref["counters"]["rays"]


def test_room_1m_full_frame_properties(room_1m):
    """The whole 3840x2160 frame at 1 spp: grouping of tiles into calls, repeat
    runs and the traversal mode."""
    import torch
    sc, _mesh, _bvh, scene = room_1m
    tl = scenes.tiles(W, H, GRID)
    px = sum(t[3] * t[4] for t in tl)
    assert px == W * H
    stream = torch.cuda.current_stream().cuda_stream
    accum = torch.empty((px * 3,), dtype=torch.float32, device="cuda")

    def frame(mode, groups):
        rgba = torch.zeros((px * 4,), dtype=torch.uint8, device="cuda")
        rays, off = 0, 0
        for g in groups:
            n = sum(t[3] * t[4] for t in g)
            st = scene.render_tiles_device([_pass(sc, t) for t in g], accum.data_ptr() + off * 12,
                                           rgba.data_ptr() + off * 4, mode=mode, stream=stream,
                                           want_stats=True)
            rays += st["rays"]
            off += n
        torch.cuda.synchronize()
        return rgba.cpu().numpy().reshape(px, 4), rays

    strict_all, rays_all = frame(api.MODE_STRICT_F32, [tl])
    # the same frame as 16 calls of one tile row each: identical bytes, counters add up
    strict_rows, rays_rows = frame(api.MODE_STRICT_F32, [tl[r * GRID:(r + 1) * GRID] for r in range(GRID)])
    assert np.array_equal(strict_all, strict_rows)
    assert rays_all == rays_rows
    # ray count bounds of Trace (trace.go:20-119): 1 primary + per vertex L shadow rays + B bounces
    assert px <= rays_all <= px * (1 + 5 * 1 + 4)
    assert (strict_all[:, 3] == 255).all()
    # ordered mode: deterministic, and equal to the reference order except for a counted few pixels
    fast_a, rays_fa = frame(api.MODE_FAST_F32, [tl])
    fast_b, rays_fb = frame(api.MODE_FAST_F32, [tl])
    assert np.array_equal(fast_a, fast_b) and rays_fa == rays_fb
    differ = int((fast_a != strict_all).any(axis=1).sum())
    record_parity_counts("config2_frame_1spp_fast_vs_strict", {
        "pixels": int(px), "pixels_differing": differ, "rays_strict": int(rays_all),
        "rays_fast": int(rays_fa), "ray_count_delta": int(rays_fa - rays_all)})
    assert differ <= px // 1000, differ
    assert abs(rays_fa - rays_all) <= px // 1000


def test_room_1m_hits_lie_in_their_triangles(room_1m, oracle):
    """Geometric check that needs no oracle: for 2M incoherent rays through the
    1M-triangle room the reported (t,u,v) reproduces the hit point inside the
    reported triangle, in both traversal orders."""
    sc, _mesh, bvh, scene = room_1m
    rng = np.random.default_rng(7)
    n = 2_000_000
    rays = np.zeros(n, api.RAY_DTYPE)
    rays["origin"] = (rng.random((n, 3), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * \
        np.array([15.0, 3.0, 15.0], np.float32) + np.array([0.0, 11.0, 0.0], np.float32)   # above the field
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays["tmax"] = np.float32(3.4028235e38)
    order = bvh.export()["order"]
    v = sc["verts"].reshape(-1, 3, 3)[order]          # triangles in array order
    for mode in (api.MODE_STRICT_F32, api.MODE_FAST_F32):
        hits, _ = scene.trace_rays(rays, mode=mode)
        ok = hits["tri_pos"] >= 0
        assert ok.mean() > 0.5
        h = hits[ok]
        tri = v[h["tri_pos"]].astype(np.float64)
        u, w_ = h["u"].astype(np.float64)[:, None], h["v"].astype(np.float64)[:, None]
        assert (h["u"] >= 0).all() and (h["v"] >= 0).all() and (h["u"] + h["v"] <= np.float32(1.0001)).all()
        on_tri = tri[:, 0] + u * (tri[:, 1] - tri[:, 0]) + w_ * (tri[:, 2] - tri[:, 0])
        on_ray = rays["origin"][ok].astype(np.float64) + rays["dir"][ok].astype(np.float64) * \
            h["t"].astype(np.float64)[:, None]
        err = np.abs(on_tri - on_ray).max(axis=1)
        assert np.quantile(err, 0.999) < 2e-3 and err.max() < 0.1, (np.quantile(err, 0.999), err.max())
    # both orders agree on all but a counted handful of the 2M rays
    a, _ = scene.trace_rays(rays, mode=api.MODE_STRICT_F32)
    b, _ = scene.trace_rays(rays, mode=api.MODE_FAST_F32)
    counts = classify_fast_differences(oracle, sc["verts"], order, rays, b, a["tri_pos"], a["t"])
    record_parity_counts("config2_scene_2M_incoherent_rays_fast_vs_strict", counts)
    assert counts["differing"] <= 40, counts
    c, _ = scene.trace_rays(rays, mode=api.MODE_FAST_NEAR_F32)      # the origins lie inside the room
    counts = classify_fast_differences(oracle, sc["verts"], order, rays, c, a["tri_pos"], a["t"])
    record_parity_counts("config2_scene_2M_incoherent_rays_fast_near_form_vs_strict", counts)
    assert counts["differing"] <= 40, counts
