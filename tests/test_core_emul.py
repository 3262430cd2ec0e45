"""The kernels' arithmetic (raytracer_amd/csrc/rt_core.h, the inline functions
the HIP kernels call) driven lane-serially on the CPU and compared bit for bit
with the oracle: traversal in both arithmetic modes, counters, and the whole
shading / bounce / resolve chain."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import api, scenes, _capi
from util import cornell, oracle_scene, product_mesh, random_rays, rays6


def emul_trace(emul, bvh, rays, mode):
    nodes, tris, _ = bvh.flatten()
    hits = np.zeros(rays.shape[0], api.HIT_DTYPE)
    cnt = np.zeros(5, np.uint64)
    emul.emul_trace(nodes.ctypes.data_as(C.c_void_p), C.c_int32(nodes.shape[0]),
                    tris.ctypes.data_as(C.c_void_p), C.c_int32(tris.shape[0]),
                    rays.ctypes.data_as(C.c_void_p), C.c_int64(rays.shape[0]), C.c_int(mode),
                    hits.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
    return hits, cnt


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,leaf,depth", [(1, 6, 16), (34, 1, 8), (5000, 6, 24), (5000, 6, 3)])
def test_traversal_bit_exact(product_lib, oracle, emul, mode, n, leaf, depth):
    sc = scenes.soup(n, seed=4242 + n)
    bvh = api.Bvh.build(product_mesh(sc), True, leaf, depth)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    rays = random_rays(6000, seed=99)
    rays["tmax"][::3] = np.float32(1.25)       # bounded (shadow-style) rays too
    hits, cnt = emul_trace(emul, bvh, rays, mode)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax, mode=mode, counters=True)
    assert np.array_equal(hits["tri_pos"], ref["tri"])
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(hits["u"].view(np.uint32), ref["u"].view(np.uint32))
    assert np.array_equal(hits["v"].view(np.uint32), ref["v"].view(np.uint32))
    c = ref["counters"]
    assert cnt.tolist() == [c["rays"], c["nodes_visited"], c["box_tests"], c["tris_fetched"],
                            c["tris_tested"]]
    assert (hits["tri_pos"] >= 0).sum() > 0 or n == 1


def test_degenerate_directions(product_lib, oracle, emul):
    """Axis-parallel rays (zero and negative-zero components -> +-Inf inverse
    directions, NaN slabs) on axis-aligned geometry."""
    sc = cornell()
    bvh = api.Bvh.build(product_mesh(sc), True, 2, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    o, d = [], []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            for k in range(40):
                org = np.array([-0.2 + 0.01 * k, 2.5, -2.0], np.float32)
                dr = np.zeros(3, np.float32)
                dr[ax] = sgn
                dr = dr * np.float32(-1) * np.float32(-1) if k % 2 else dr * np.float32(1)
                if k % 3 == 0:
                    dr = (dr * np.float32(-1)) * np.float32(-1)
                o.append(org)
                d.append(-dr if k % 4 == 0 else dr)
    rays = api.make_rays(np.array(o), np.array(d))
    hits, _ = emul_trace(emul, bvh, rays, 0)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax)
    assert np.array_equal(hits["tri_pos"], ref["tri"])
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert (hits["tri_pos"] >= 0).sum() > 100


def _pass_params(light, p):
    buf = (C.c_char * 512)()
    _capi.check(_capi.lib().rt_pass_params(C.byref(light), C.byref(p), buf, 512))
    return buf


def emul_render(emul, mesh, bvh, sc, p, cam_tab, light_tab, sample_begin, sample_count):
    nodes, tris, mat = bvh.flatten()
    light = mesh.light(bvh)
    buf = _pass_params(light, p)
    pixels = p.width * p.height
    accum = np.zeros((pixels, 3), np.float32)
    rad = np.zeros((sample_count, pixels, 3), np.float32)
    rays = C.c_uint64()
    kd = np.ascontiguousarray(sc["kd"], np.float32)
    emul.emul_render(nodes.ctypes.data_as(C.c_void_p), C.c_int32(nodes.shape[0]),
                     tris.ctypes.data_as(C.c_void_p), C.c_int32(tris.shape[0]),
                     mat.ctypes.data_as(C.c_void_p), kd.ctypes.data_as(C.c_void_p),
                     cam_tab.ctypes.data_as(C.c_void_p), C.c_int32(cam_tab.shape[0]),
                     light_tab.ctypes.data_as(C.c_void_p), C.c_int32(light_tab.shape[0]),
                     buf, C.c_int32(sample_begin), C.c_int32(sample_count),
                     accum.ctypes.data_as(C.c_void_p), rad.ctypes.data_as(C.c_void_p),
                     C.byref(rays))
    rgba = np.zeros((pixels, 4), np.uint8)
    emul.emul_resolve(accum.ctypes.data_as(C.c_void_p), C.c_int64(pixels),
                      C.c_int32(sample_begin + sample_count), C.c_int32(p.settings.gamma_correction),
                      C.c_float(p.settings.gamma), rgba.ctypes.data_as(C.c_void_p))
    return accum, rad, rgba, rays.value


CORNELL_CAM = {"x": -0.225, "y": 2.55, "z": 6.0}


@pytest.mark.parametrize("L,B,spp", [(6, 3, 2), (1, 0, 1), (2, 5, 1)])
def test_cornell_render_bit_exact(product_lib, oracle, emul, L, B, spp):
    sc = cornell()
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    cam_tab = api.default_sample_table(12345, 1)
    light_tab = api.default_sample_table(100001, 2)
    osc.set_sample_tables(cam_tab, light_tab)
    d = scenes.make_pass_dict(40, 40, CORNELL_CAM, spp=spp, bounces=B, light_samples=L,
                              intensity=100.0, seed=7, x_offset=10, y_offset=20, tile_w=24,
                              tile_h=16, task_id=3)
    p = api.pass_from_dict(d)
    po = oracle.pass_from_dict(d)
    ref = osc.render_tile(po, want_radiance=True)
    accum, rad, rgba, rays = emul_render(emul, mesh, bvh, sc, p, cam_tab, light_tab, 0, spp)
    assert np.array_equal(rad.view(np.uint32).reshape(-1), ref["radiance"].view(np.uint32).reshape(-1))
    assert np.array_equal(accum.view(np.uint32).reshape(-1), ref["accum"].view(np.uint32).reshape(-1))
    assert np.array_equal(rgba.reshape(-1), ref["rgba"].reshape(-1))
    assert rays == ref["counters"]["rays"]
    assert ref["rgba"][..., :3].max() > 50          # the image is lit
    assert rays > 24 * 16 * spp


def test_light_and_camera_setup_match(product_lib, oracle):
    sc = cornell()
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    a, b = mesh.light(bvh), osc.light()
    assert a.found == 1 and b.found == 1
    assert list(a.transform) == list(b.transform) and list(a.size) == list(b.size)
    assert list(a.normal) == list(b.normal)
    # the Cornell light faces down and is about 0.525 x 0.65 (half sizes)
    assert a.normal[1] < -0.99 and abs(a.size[0] - 0.525) < 1e-3 and abs(a.size[1] - 0.65) < 1e-3
    for proj, w, h, fov in [(0, 250, 250, 45.0), (0, 1920, 1080, 60.0), (1, 640, 480, 45.0)]:
        d = scenes.make_pass_dict(w, h, CORNELL_CAM, fov=fov, projection=proj)
        tl, st = api.camera_setup(api.pass_from_dict(d).camera, w, h)
        tl2, st2 = oracle.camera_setup(oracle.pass_from_dict(d).camera, w, h)
        assert np.array_equal(tl, tl2) and np.array_equal(st, st2)
        assert st[0] > 0 and st[1] > 0


def test_debug_light_when_scene_has_none(product_lib, oracle, emul):
    """No "Light" material -> RenderPass.Initialize makes a debug light at the
    camera or at the given transform (context.go:257-271)."""
    sc = scenes.soup(300, seed=8)
    mesh = product_mesh(sc)
    bvh = api.Bvh.build(mesh, True, 6, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    cam_tab = api.default_sample_table(97, 1)
    light_tab = api.default_sample_table(211, 2)
    osc.set_sample_tables(cam_tab, light_tab)
    d = scenes.make_pass_dict(32, 24, {"x": 0, "y": 0, "z": 4}, spp=1, bounces=2, light_samples=2,
                              intensity=30.0, seed=3, gamma_on=False)
    d["Settings"]["DebugLightAtCamera"] = True
    d["Settings"]["DebugLightSize"] = 0.5
    ref = osc.render_tile(oracle.pass_from_dict(d), want_radiance=True)
    accum, rad, rgba, rays = emul_render(emul, mesh, bvh, sc, api.pass_from_dict(d), cam_tab,
                                         light_tab, 0, 1)
    assert np.array_equal(accum.view(np.uint32).reshape(-1), ref["accum"].view(np.uint32).reshape(-1))
    assert np.array_equal(rgba.reshape(-1), ref["rgba"].reshape(-1))


def test_default_sample_table_is_low_discrepancy(product_lib):
    t = api.default_sample_table(4096, 5)
    assert t.min() >= 0 and t.max() < 1
    # every cell of a 16x16 grid is hit by a scrambled Halton prefix of 4096
    cells = np.unique((t[:, 0] * 16).astype(int) * 16 + (t[:, 1] * 16).astype(int))
    assert len(cells) == 256
    assert abs(t.mean() - 0.5) < 0.01
    assert not np.array_equal(t, api.default_sample_table(4096, 6))


def emul_trace_fast(emul, bvh, rays, near=False):
    rec, root = bvh.flatten_ordered()
    _, tris, _ = bvh.flatten()
    hits = np.zeros(rays.shape[0], api.HIT_DTYPE)
    cnt = np.zeros(5, np.uint64)
    emul.emul_trace_fast(rec.ctypes.data_as(C.c_void_p), C.c_uint32(root),
                         tris.ctypes.data_as(C.c_void_p), C.c_int32(tris.shape[0]),
                         rays.ctypes.data_as(C.c_void_p), C.c_int64(rays.shape[0]),
                         hits.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), C.c_int(int(near)))
    return hits, cnt


@pytest.mark.parametrize("near", [False, True])
@pytest.mark.parametrize("n,leaf,depth", [(1, 6, 16), (34, 1, 8), (5000, 6, 24), (400, 200, 4),
                                          (3000, 1, 12)])
def test_ordered_traversal_finds_the_same_hits(product_lib, oracle, emul, n, leaf, depth, near):
    """Fast mode: same closest hit as the reference order on generic rays (no
    exact ties in a random soup), with far less work; in both forms of the step
    (the rays start inside the near region: the soup's box grown by its diagonal)."""
    sc = scenes.soup(n, seed=777 + n)
    bvh = api.Bvh.build(product_mesh(sc), True, leaf, depth)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    rays = random_rays(8000, seed=5)
    rays["tmax"][::4] = np.float32(1.5)
    hits, cnt = emul_trace_fast(emul, bvh, rays, near)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax, counters=True)
    diff = hits["tri_pos"] != ref["tri"]
    assert diff.sum() == 0, int(diff.sum())
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(hits["u"].view(np.uint32), ref["u"].view(np.uint32))
    if n >= 3000:
        # less work in the byte model of the traversal (56 B per interior record, 48 B per triangle)
        c = ref["counters"]
        assert 56 * int(cnt[1]) + 48 * int(cnt[3]) < 56 * c["nodes_visited"] + 48 * c["tris_fetched"]


def test_ordered_traversal_with_point_triangles(product_lib, oracle, emul):
    """Records whose children are all one point (triangles collapsed to a point,
    some of them identical) have no extent to quantise; the unused slots of such
    records must still keep rays out, or at least cost nothing but work."""
    sc = scenes.soup(600, seed=31)
    v = sc["verts"].reshape(-1, 3, 3).copy()
    v[100:140] = v[100:140, :1, :]                       # 40 point triangles
    v[140:150] = v[140, 0][None, None, :]                # 10 of them on one point
    v[150:153] = np.float32(0.0)                         # and 3 at the origin
    sc = dict(sc, verts=np.ascontiguousarray(v.reshape(-1, 9)))
    bvh = api.Bvh.build(product_mesh(sc), True, 2, 24)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    rays = random_rays(6000, seed=9)
    rays["origin"][::3] *= np.float32(1e6)               # far origins: extents below rounding
    hits, cnt = emul_trace_fast(emul, bvh, rays)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax)
    assert np.array_equal(hits["tri_pos"], ref["tri"])
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    rec, _ = bvh.flatten_ordered()
    assert np.isfinite(rec[:, 0:6]).all() and (rec[:, 3:6] > 0).all()


def _decode_wide(rec):
    """Child boxes of 64-byte ordered-traversal records: (n, 4, 3) lower and
    upper planes, decoded like the traversal does, fma(offset, step, origin)."""
    origin = rec[:, 0:3].astype(np.float64)
    step = rec[:, 3:6].astype(np.float64)
    planes = rec[:, 6:12].view(np.uint32)
    k = np.arange(4, dtype=np.uint32)[None, :, None]
    lo_q = (planes[:, None, 0:3] >> (8 * k)) & 0xFF
    hi_q = (planes[:, None, 3:6] >> (8 * k)) & 0xFF
    # offset * step is exact in float64 (8 bits x 24 bits), so rounding the sum once to
    # float32 equals the fused multiply-add
    lo = (lo_q * step[:, None, :] + origin[:, None, :]).astype(np.float32)
    hi = (hi_q * step[:, None, :] + origin[:, None, :]).astype(np.float32)
    return lo, hi


def test_ordered_records_layout(product_lib):
    sc = scenes.soup(300, seed=2)
    bvh = api.Bvh.build(product_mesh(sc), True, 6, 16)
    e = bvh.export()
    rec, root = bvh.flatten_ordered()
    assert rec.shape[1] == 16 and root == 0
    refs = rec[:, 12:16].view(np.uint32)
    unused = refs == 0x80000000                       # empty slot: leaf bit, count 0
    leaf = ((refs & 0x80000000) != 0) & ~unused
    counts = (refs[leaf] >> 24) & 0x7F
    starts = refs[leaf] & 0xFFFFFF
    # leaves of more than 4 triangles are cut in two by array position (smaller ones
    # where the builder finds it cheaper): the small leaves tile the array and
    # every reference leaf start is among theirs
    assert counts.sum() == 300 and counts.max() <= 4 and counts.min() >= 1
    order = np.argsort(starts)
    assert np.array_equal(np.cumsum(counts[order]) - counts[order], starts[order])
    assert set(e["start"][e["is_leaf"] == 1].tolist()) <= set(starts.tolist())
    interior = ~leaf & ~unused
    assert np.all(refs[interior] < rec.shape[0])
    # every record but the root is referenced exactly once; used slots come first
    assert sorted(refs[interior].tolist()) == list(range(1, rec.shape[0]))
    assert (np.diff(unused.astype(int), axis=1) >= 0).all()
    # an unused slot carries an inverted box (lower planes at offset 255, upper at 0),
    # and its record has an axis on which offset 255 decodes above offset 0: the slab
    # interval of that box is empty, which is all that keeps rays out of the slot
    planes = rec[:, 6:12].view(np.uint32)
    k = np.arange(4, dtype=np.uint32)[None, :, None]
    lo_q = (planes[:, None, 0:3] >> (8 * k)) & 0xFF
    hi_q = (planes[:, None, 3:6] >> (8 * k)) & 0xFF
    assert unused.any()
    assert (lo_q[unused] == 255).all() and (hi_q[unused] == 0).all()
    top = (255.0 * rec[:, 3:6].astype(np.float64) + rec[:, 0:3].astype(np.float64)).astype(np.float32)
    assert ((top > rec[:, 0:3]).any(axis=1))[unused.any(axis=1)].all()
    # steps are positive and finite; decoded leaf boxes contain their triangles
    step = rec[:, 3:6]
    assert np.isfinite(step).all() and (step > 0).all()
    lo, hi = _decode_wide(rec)
    ext = e["bmax"][0].astype(np.float64) - e["bmin"][0]
    margin = (ext + np.linalg.norm(ext)) * 2.0 ** -23          # rt_core.h: kNearOriginDiagonals = 1
    v = sc["verts"].reshape(-1, 3, 3)[e["order"]]
    for r, c in zip(*np.nonzero(leaf)):
        s0, n0 = int(refs[r, c] & 0xFFFFFF), int((refs[r, c] >> 24) & 0x7F)
        tri = v[s0:s0 + n0].reshape(-1, 3)
        assert (lo[r, c] <= tri.min(axis=0)).all() and (hi[r, c] >= tri.max(axis=0)).all()
        # ... with the margin the near-origin form of the step needs (2^-23 of the farthest a
        # qualifying ray origin can be from a frame origin, per axis) ...
        # (in exact arithmetic, offset * step + origin before any rounding: what the near-origin
        # form adds the ray-relative frame origin to)
        x_lo = lo_q[r, c] * step[r].astype(np.float64) + rec[r, 0:3].astype(np.float64)
        x_hi = hi_q[r, c] * step[r].astype(np.float64) + rec[r, 0:3].astype(np.float64)
        assert (tri.min(axis=0).astype(np.float64) - x_lo >= margin * 0.999).all()
        assert (x_hi - tri.max(axis=0).astype(np.float64) >= margin * 0.999).all()
        # ... and are not looser than one step plus that margin per side
        assert (tri.min(axis=0) - lo[r, c] <= step[r] * 1.001 + margin).all()
        assert (hi[r, c] - tri.max(axis=0) <= step[r] * 1.001 + margin).all()
    # the root record's children cover the root box of the reference tree
    used0 = ~unused[0]
    assert (lo[0][used0].min(axis=0) <= e["bmin"][0]).all() and (hi[0][used0].max(axis=0) >= e["bmax"][0]).all()
    assert (rec[0, 0:3] <= e["bmin"][0]).all() and (e["bmin"][0] - rec[0, 0:3] <= margin * 1.01 + 1e-7).all()
    # a tree that is one big leaf becomes a tree of small leaves
    big = api.Bvh.build(product_mesh(scenes.soup(500, seed=3)), False, 6, 16)
    rec, root = big.flatten_ordered()
    refs = rec[:, 12:16].view(np.uint32)
    leaf = ((refs & 0x80000000) != 0) & (refs != 0x80000000)
    assert (root & 0x80000000) == 0 and ((refs[leaf] >> 24) & 0x7F).sum() == 500
    assert (((refs[leaf] >> 24) & 0x7F) <= 4).all()


def test_ordered_records_lead_with_the_largest_frames(product_lib):
    """The flattener puts the records with the largest frames (255-step boxes) first,
    largest first, because the ordered kernel keeps a prefix of the array in LDS
    (flatten.cpp: top_records_first); the others stay in depth-first order, a record's
    first interior child right behind it."""
    sc = scenes.soup(9000, seed=7)
    bvh = api.Bvh.build(product_mesh(sc), True, 6, 32)
    rec, root = bvh.flatten_ordered()
    n = rec.shape[0]
    assert root == 0 and n > 1024 + 100
    ext = 255.0 * rec[:, 3:6].astype(np.float64)
    area = ext[:, 0] * ext[:, 1] + ext[:, 0] * ext[:, 2] + ext[:, 1] * ext[:, 2]
    top = 1024                                            # rt_core.h: kWideTopRecords
    assert (np.diff(area[:top]) <= 0).all()
    assert area[:top].min() >= area[top:].max()
    refs = rec[:, 12:16].view(np.uint32)
    interior = (refs & 0x80000000) == 0
    # still one tree: every record but the root referenced exactly once
    assert sorted(refs[interior].tolist()) == list(range(1, n))
    # a parent's frame contains its children's frames, so the front is a top of the tree:
    # every record of the front is referenced from the front
    parent = np.zeros(n, np.int64)
    rows = np.nonzero(interior)[0]
    parent[refs[interior]] = rows
    assert (parent[1:top] < top).all()
    # behind the front: depth-first order, i.e. a record's first interior child that is
    # not in the front follows it directly whenever the record itself is not in the front
    deep = np.arange(top, n)
    first_child = np.array([next((int(r) for r, i in zip(refs[k], interior[k]) if i and r >= top), -1) for k in deep])
    has = first_child >= 0
    assert (first_child[has] == deep[has] + 1).all()


def test_textured_render_bit_exact(product_lib, oracle, emul):
    """map_Kd sampling of the device functions (material_diffuse) against the
    oracle's getMaterialParameters: repeat wrap incl. negative coordinates, odd
    texture sizes, textured and untextured materials side by side."""
    sc = scenes.room_mesh(24, seed=0x5EED0002, total=1400)
    n = sc["verts"].shape[0]
    rng = np.random.default_rng(3)
    uv = (rng.random((n, 6)) * 5.0 - 2.0).astype(np.float32)
    tex_of_mat = np.full(sc["kd"].shape[0], -1, np.int32)
    tex_of_mat[0], tex_of_mat[2] = 1, 0
    images = [np.ascontiguousarray(rng.integers(0, 256, (9, 17, 4), dtype=np.uint8)),
              np.ascontiguousarray(rng.integers(0, 256, (7, 5, 4), dtype=np.uint8))]
    mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"],
                    texcoords=uv, texture=tex_of_mat)
    bvh = api.Bvh.build(mesh, True, 6, 24)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    cam_tab = api.default_sample_table(12345, 1)
    light_tab = api.default_sample_table(100001, 2)
    osc.set_sample_tables(cam_tab, light_tab)
    osc.set_texcoords(uv)
    for m, slot in enumerate(tex_of_mat):
        if slot >= 0:
            osc.set_texture(m, images[slot])
    d = scenes.make_pass_dict(48, 32, sc["camera"], spp=2, bounces=2, light_samples=2,
                              intensity=6.0, seed=9)
    ref = osc.render_tile(oracle.pass_from_dict(d), want_radiance=True)
    order = bvh.export()["order"]
    tri_uv = np.ascontiguousarray(uv[order])                 # array order, like the flattened triangles
    ptrs = (C.c_void_p * len(images))(*[im.ctypes.data for im in images])
    dims = np.array([[im.shape[1], im.shape[0]] for im in images], np.int32)
    emul.emul_set_textures(tri_uv.ctypes.data_as(C.c_void_p), tex_of_mat.ctypes.data_as(C.c_void_p),
                           ptrs, dims.ctypes.data_as(C.c_void_p))
    try:
        accum, rad, rgba, rays = emul_render(emul, mesh, bvh, sc, api.pass_from_dict(d), cam_tab,
                                             light_tab, 0, 2)
    finally:
        emul.emul_set_textures(None, None, None, None)
    assert np.array_equal(accum.view(np.uint32).reshape(-1), ref["accum"].view(np.uint32).reshape(-1))
    assert np.array_equal(rgba.reshape(-1), ref["rgba"].reshape(-1))
    assert rays == ref["counters"]["rays"]


@pytest.mark.parametrize("scale,offset", [(1.0e6, 0.0), (0.6, 0.0), (1.0, 4.0e4), (1.0e3, -7.0e6), (1.0e-3, 1.0e3)])
def test_ordered_traversal_at_extreme_scales(product_lib, oracle, emul, scale, offset):
    """The 8-bit child boxes of the ordered view live in a per-record frame.
    Scenes that are huge, tiny or far from the origin (one frame step = many
    float32 ulps, or less than one ulp of the coordinates) must keep every
    decoded box around the reference's box: same hits as the reference order."""
    sc = scenes.soup(3000, seed=77, extent=1.0, size=0.08)
    sc["verts"] = (sc["verts"] * np.float32(scale) + np.float32(offset)).astype(np.float32)
    bvh = api.Bvh.build(product_mesh(sc), True, 6, 32)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    rays = random_rays(12000, seed=4, extent=1.2)
    rays["origin"] = (rays["origin"] * np.float32(scale) + np.float32(offset)).astype(np.float32)
    rays["dir"][::9, 0] = np.float32(1e-41)
    rays["dir"][4::9, 2] = np.float32(0.0)
    hits, _ = emul_trace_fast(emul, bvh, rays)
    r6, tmax = rays6(rays)
    ref = osc.trace_rays(r6, tmax)
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
    # decoded leaf boxes contain their triangles at every scale
    rec, _root = bvh.flatten_ordered()
    refs = rec[:, 12:16].view(np.uint32)
    leaf = ((refs & 0x80000000) != 0) & (refs != 0x80000000)
    lo, hi = _decode_wide(rec)
    v = sc["verts"].reshape(-1, 3, 3)[bvh.export()["order"]]
    for r, c in zip(*np.nonzero(leaf)):
        s0, n0 = int(refs[r, c] & 0xFFFFFF), int((refs[r, c] >> 24) & 0x7F)
        tri = v[s0:s0 + n0].reshape(-1, 3)
        assert (lo[r, c] <= tri.min(axis=0)).all() and (hi[r, c] >= tri.max(axis=0)).all()
