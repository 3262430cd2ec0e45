"""The reference's entry points (main.go:28-33) through the native backend:
same names, JSON in, JSON out.  CPU part: ingestion, BVH hand-over, error
documents, PNG decoding.  GPU part: the bundled Cornell-box preset rendered
tile by tile exactly as the frontend schedules it, against the oracle."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from raytracer_amd import api, backend, scenes
from util import GOLDEN, cornell, oracle_scene, product_mesh

CORNELL_DIR = os.path.join(GOLDEN, "cornell-box")
# frontend/app/public/presets/cornell-box.json, the values the reference's demo uses
PRESET = {"width": 250, "height": 250, "x": -0.225, "y": 2.55, "z": 6, "rx": 0, "ry": 0, "rz": 0,
          "projection": 0, "projectionPlaneDistance": 1, "fieldOfView": 45, "bounces": 3,
          "lightSampleRays": 6, "raysPerPixel": 25, "taskCount": 16, "gammaCorrection": True,
          "gamma": 2.2, "useBVH": True, "maxLeafSize": 6, "maxDepth": 16, "lightIntensity": 100}


def _context_json(**over):
    with open(os.path.join(CORNELL_DIR, "cornell-box.obj")) as f:
        obj = f.read()
    with open(os.path.join(CORNELL_DIR, "cornell-box.mtl")) as f:
        mtl = f.read()
    ctx = {"Debug": False, "UseBVH": PRESET["useBVH"], "BVHMaxLeafSize": PRESET["maxLeafSize"],
           "BVHMaxDepth": PRESET["maxDepth"], "Scene": {}, "ObjBuffer": obj, "MtlBuffer": mtl,
           "RawTextures": [], "WorkerId": 0}
    ctx.update(over)
    return json.dumps(ctx)


def _tasks(spp, seed=1):
    """Renderer.js:356-443: sqrt(taskCount)^2 tiles, RNGSeed = base + TaskID."""
    n = int(np.sqrt(PRESET["taskCount"]))
    cam = {"x": PRESET["x"], "y": PRESET["y"], "z": PRESET["z"]}
    out = []
    for tid, x, y, w, h in scenes.tiles(PRESET["width"], PRESET["height"], n):
        out.append(scenes.make_pass_dict(PRESET["width"], PRESET["height"], cam, spp=spp,
                                         bounces=PRESET["bounces"], light_samples=PRESET["lightSampleRays"],
                                         gamma=PRESET["gamma"], intensity=PRESET["lightIntensity"],
                                         seed=seed, x_offset=x, y_offset=y, tile_w=w, tile_h=h,
                                         task_id=tid))
    return out


def test_initialize_build_load_round_trip(product_lib):
    events = []
    b = backend.Backend(progressUpdate=events.append)
    assert b.triangle_count == 0
    b.initialize(_context_json())
    assert b.triangle_count == 34
    text = b.buildBVH()
    doc = json.loads(text)
    assert doc["Root"]["StartIndex"] == 0 and doc["Root"]["EndIndex"] == 33
    b.loadBVH(text)
    # same tree as the array-level API on the same triangles
    sc = cornell()
    ref = api.Bvh.build(product_mesh(sc), True, 6, 16)
    assert json.loads(ref.to_json()) == doc
    names = [json.loads(e)["event"] for e in events]
    assert names == ["RenderContext.Initialize"] * 2 + ["RenderContext.BuildBVH"] * 2 + ["RenderContext.LoadBVH"] * 2
    assert set(json.loads(events[0])) == {"progress", "event", "taskId", "rays"}
    assert json.loads(events[0])["taskId"] == -1


def test_obj_ingestion_rules(product_lib):
    """Groups restart Triangle.Index, quads split, relative indices, unknown
    material -> Debug (context.go:109-168)."""
    obj = "\n".join([
        "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0 0 1", "vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1",
        "usemtl A", "f 1/1 2/2 3/3 4/4", "f -5 -4 -3",
        "usemtl Missing", "f 1 2 5",
        "usemtl A", "f 1//1 3//1 5//1"])
    mtl = "newmtl A\nKd 0.1 0.2 0.3\nnewmtl Light\nKd 1 1 1\n"
    b = backend.Backend()
    b.initialize(json.dumps({"UseBVH": False, "BVHMaxLeafSize": 6, "BVHMaxDepth": 4,
                             "ObjBuffer": obj, "MtlBuffer": mtl}))
    assert b.triangle_count == 5          # quad -> 2, three triangles
    doc = json.loads(b.buildBVH())
    assert doc["Root"]["LeftChild"] is None and doc["Root"]["EndIndex"] == 4
    assert doc["Root"]["Bounds"]["Min"] == [0, 0, 0] and doc["Root"]["Bounds"]["Max"] == [1, 1, 1]
    with pytest.raises(api._capi.RtError) as e:
        b.initialize(json.dumps({"ObjBuffer": "v 0 0 0\nf 1 2 3\n", "MtlBuffer": "newmtl A\n"}))
    assert e.value.status == -4 and "line 2" in e.value.message
    with pytest.raises(api._capi.RtError):
        b.initialize(json.dumps({"ObjBuffer": "", "MtlBuffer": ""}))
    with pytest.raises(api._capi.RtError):
        b.initialize("{not json")


def test_render_errors_come_back_as_documents(product_lib):
    """handleError (main.go:308-325): exitCode -1, message, blank 500x500."""
    b = backend.Backend()
    b.initialize(_context_json())
    for bad in ("{", json.dumps({"Width": "wide"}), json.dumps(_tasks(1)[0])):   # last: no BVH loaded
        doc = json.loads(b.render(bad))
        assert doc["exitCode"] == -1 and doc["message"]
        img = backend.decode_image(doc)
        assert img.shape == (500, 500, 4) and not img.any()
    doc = json.loads(b.incrementalRender())
    assert doc["exitCode"] == -1 and "initializeIncrementalRender" in doc["message"]
    # encoding/json refuses fractions and out-of-range values for integer fields
    good = _tasks(1)[0]
    for field, value in (("Width", 62.5), ("TaskID", 1e30), ("XOffset", -3e9), ("RNGSeed", 0.5)):
        doc = json.loads(b.render(json.dumps(dict(good, **{field: value}))))
        assert doc["exitCode"] == -1 and field in doc["message"], (field, doc["message"])
    for field, value in (("BounceLimit", 256), ("BounceLimit", 1.5), ("LightSampleRays", 2.5)):
        bad = dict(good, Settings=dict(good["Settings"], **{field: value}))
        doc = json.loads(b.render(json.dumps(bad)))
        assert doc["exitCode"] == -1 and field in doc["message"], (field, doc["message"])
    with pytest.raises(api._capi.RtError):
        b.initialize(_context_json(BVHMaxDepth=16.5))


def _png(w, h, ctype, rows, palette=None, trns=None, depth=8):
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    raw = b"".join(b"\x00" + bytes(r) for r in rows)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    return out + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")


def test_png_decoding(product_lib):
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    out = api.png_decode(_png(17, 13, 2, [r.reshape(-1) for r in rgb]))
    assert np.array_equal(out[..., :3], rgb) and (out[..., 3] == 255).all()
    rgba = rng.integers(0, 256, (9, 5, 4), dtype=np.uint8)
    rgba[0, 0] = (200, 100, 50, 255)
    rgba[0, 1] = (200, 100, 50, 0)
    out = api.png_decode(_png(5, 9, 6, [r.reshape(-1) for r in rgba]))
    a = rgba[..., 3].astype(np.uint32)
    opaque = a == 255
    for ch in range(3):     # image/draw: (c * a*0x101 / 0xff) >> 8
        pre = ((rgba[..., ch].astype(np.uint32) * (a * 0x101) // 0xff) >> 8).astype(np.uint8)
        expect = np.where(opaque, rgba[..., ch], pre)
        assert np.array_equal(out[..., ch], expect)
    assert np.array_equal(out[..., 3], rgba[..., 3])
    # a large smooth image makes zlib emit dynamic-Huffman blocks with long matches
    yy, xx = np.mgrid[0:128, 0:256]
    big = np.stack([xx % 256, (yy * 2) % 256, (xx // 4 + yy) % 256], axis=-1).astype(np.uint8)
    assert np.array_equal(api.png_decode(_png(256, 128, 2, [r.reshape(-1) for r in big]))[..., :3], big)
    # palette image with 2-bit indices
    pal = [255, 0, 0, 0, 255, 0, 0, 0, 255, 9, 9, 9]
    rows = [[0b00011011, 0b11000000], [0b10100101, 0b01000000]]
    out = api.png_decode(_png(5, 2, 3, rows, palette=pal, depth=2))
    assert out[0, :5, :3].tolist() == [[255, 0, 0], [0, 255, 0], [0, 0, 255], [9, 9, 9], [9, 9, 9]]
    with pytest.raises(api._capi.RtError):
        api.png_decode(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    with pytest.raises(api._capi.RtError):
        api.png_decode(b"GIF89a" + b"\0" * 40)
    # a header that promises far more pixels than the data stream can hold is refused
    # before anything of that size is allocated; extra data after the image is ignored
    tiny = _png(4, 4, 2, [[7] * 12] * 4)
    huge = tiny[:16] + struct.pack(">II", 60000, 60000) + tiny[24:29]
    huge += struct.pack(">I", zlib.crc32(huge[12:29])) + tiny[33:]
    with pytest.raises(api._capi.RtError):
        api.png_decode(huge)
    assert (api.png_decode(_png(4, 4, 2, [[7] * 12] * 9))[..., :3] == 7).all()


@pytest.mark.gpu
def test_cornell_preset_tile_by_tile_against_oracle(product_lib, oracle, gpu_device):
    """The reference's demo workload (preset cornell-box.json) at 3 spp: every
    one of the 16 tiles, rendered through render(passJSON), equals the oracle's
    main.go render() of the same pass, and the rays counter matches."""
    events = []
    b = backend.Backend(device=gpu_device, progressUpdate=events.append)
    b.initialize(_context_json())
    b.loadBVH(b.buildBVH())
    sc = cornell()
    bvh = api.Bvh.build(product_mesh(sc), True, 6, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    osc.set_sample_tables(api.default_sample_table(12345, 0xC0FFEE01),
                          api.default_sample_table(100001, 0xC0FFEE02))
    frame = np.zeros((250, 250, 4), np.uint8)
    total_rays = 0
    for d in _tasks(spp=3):
        doc = json.loads(b.render(json.dumps(d)))
        assert doc["exitCode"] == 0 and doc["message"] == ""
        img = backend.decode_image(doc)
        assert doc["imageData"]["Stride"] == 4 * d["Width"]
        ref = osc.render_tile(oracle.pass_from_dict(d))
        assert np.array_equal(img, ref["rgba"]), d["TaskID"]
        total_rays += ref["counters"]["rays"]
        frame[d["YOffset"]:d["YOffset"] + d["Height"], d["XOffset"]:d["XOffset"] + d["Width"]] = img
    assert b.rays == total_rays
    assert frame[..., :3].mean() > 40 and (frame[..., 3] == 255).all()
    trace_events = [json.loads(e) for e in events if json.loads(e)["event"] == "trace"]
    # per tile: 0.0, nine intermediate updates, 1.0 (main.go:120-154)
    assert len(trace_events) == 16 * 11 and trace_events[-1]["rays"] == total_rays
    assert all(a["rays"] <= b["rays"] for a, b in zip(trace_events, trace_events[1:]))
    # a new RenderKey resets the counter (main.go:107-110)
    d = _tasks(spp=1)[0]
    d["RenderKey"] = 2
    b.render(json.dumps(d))
    assert 0 < b.rays < total_rays


@pytest.mark.gpu
def test_render_progress_events_follow_the_reference(product_lib, oracle, gpu_device):
    """One 62x62 tile of the preset at its own 25 spp, 6 light samples, 3
    bounces: the progressUpdate documents are the ones main.go:120-156,185
    issues -- trace 0.0, an update whenever the running sample index passes
    updateIndex + updateInterval (progress = updateIndex / rayCount, rays = the
    counter's value at that sample), trace 1.0, output 0.0, output 1.0 -- with
    the oracle's running ray counts, continued from the counter of earlier
    calls with the same RenderKey."""
    events = []
    b = backend.Backend(device=gpu_device, progressUpdate=events.append)
    b.initialize(_context_json())
    b.loadBVH(b.buildBVH())
    sc = cornell()
    bvh = api.Bvh.build(product_mesh(sc), True, 6, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    osc.set_sample_tables(api.default_sample_table(12345, 0xC0FFEE01),
                          api.default_sample_table(100001, 0xC0FFEE02))
    tasks = _tasks(spp=25)
    base = 0
    for d in (tasks[5], tasks[15]):            # a 62x62 tile, then the 64x64 corner tile
        del events[:]
        doc = json.loads(b.render(json.dumps(d)))
        assert doc["exitCode"] == 0
        want = oracle.render_progress(osc, oracle.pass_from_dict(d), base_rays=base)
        assert len(want) == 11
        got = [json.loads(e) for e in events]
        assert [e["event"] for e in got] == ["trace"] * 11 + ["output"] * 2
        assert all(e["taskId"] == d["TaskID"] for e in got)
        for e, (progress, rays) in zip(got[:11], want):
            assert np.float32(e["progress"]) == np.float32(progress), (e, progress)
            assert e["rays"] == rays, (e, rays)
        assert [e["progress"] for e in got[11:]] == [0, 1]
        assert got[11]["rays"] == got[12]["rays"] == want[-1][1] == b.rays
        base = b.rays
    assert d["Width"] == 64 and tasks[5]["Width"] == 62


@pytest.mark.gpu
def test_incremental_render_flow(product_lib, oracle, gpu_device):
    b = backend.Backend(device=gpu_device)
    b.initialize(_context_json())
    b.loadBVH(b.buildBVH())
    sc = cornell()
    bvh = api.Bvh.build(product_mesh(sc), True, 6, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    osc.set_sample_tables(api.default_sample_table(12345, 0xC0FFEE01),
                          api.default_sample_table(100001, 0xC0FFEE02))
    d = _tasks(spp=3)[5]
    assert b.initializeIncrementalRender(json.dumps(d)) is None
    acc = np.zeros((d["Height"], d["Width"], 3), np.float32)
    po = oracle.pass_from_dict(d)
    for s in range(3):
        img = backend.decode_image(b.incrementalRender())
        ref = osc.render_tile(po, sample_begin=s, sample_count=1, incremental=True, accum=acc)
        assert np.array_equal(img, ref["rgba"]), s
    assert np.array_equal(b.last_image(), img)


@pytest.mark.gpu
def test_textured_material(product_lib, oracle, gpu_device):
    """map_Kd lookups (trace.go:161-181, texture.go:48-59): nearest texel, no V
    flip, repeat wrap; checked against a numpy evaluation of the primary hits."""
    obj = "\n".join(["v -2 -2 0", "v 2 -2 0", "v 2 2 0", "v -2 2 0",
                     "vt 0 0", "vt 2 0", "vt 2 2", "vt 0 2",
                     "usemtl Tex", "f 1/1 2/2 3/3", "f 1/1 3/3 4/4"])
    mtl = "newmtl Tex\nKd 1 1 1\nmap_Kd checker.png\n"
    tex = np.zeros((4, 4, 3), np.uint8)
    tex[:, :] = (0, 255, 0)
    tex[::2, ::2] = (255, 0, 0)
    tex[1::2, 1::2] = (0, 0, 255)
    png = _png(4, 4, 2, [r.reshape(-1) for r in tex])
    b = backend.Backend(device=gpu_device)
    b.initialize(json.dumps({"UseBVH": True, "BVHMaxLeafSize": 1, "BVHMaxDepth": 4, "ObjBuffer": obj,
                             "MtlBuffer": mtl, "RawTextures": [{"Name": "checker.png"}]}), png)
    b.loadBVH(b.buildBVH())
    d = scenes.make_pass_dict(64, 64, {"x": 0, "y": 0, "z": 4}, spp=1, bounces=0, light_samples=4,
                              intensity=40.0, seed=1, gamma_on=False)
    d["Settings"]["DebugLightAtCamera"] = True
    d["Settings"]["DebugLightSize"] = 0.05
    doc = json.loads(b.render(json.dumps(d)))
    assert doc["exitCode"] == 0, doc["message"]
    img = backend.decode_image(doc)[..., :3].astype(int)
    # the quad fills the view; texel colour decides which channels are lit
    lit = img.sum(axis=2) > 0
    assert lit.mean() > 0.9
    dominant = np.argmax(img, axis=2)
    only_one_channel = ((img > 0).sum(axis=2) <= 1)
    assert only_one_channel[lit].mean() > 0.99
    # all three texel colours appear, in a repeating pattern (uv spans [0,2))
    assert {0, 1, 2} <= set(np.unique(dominant[lit]).tolist())
    black = (~lit).mean()
    assert black < 0.1


@pytest.mark.gpu
def test_progress_events_arrive_while_a_large_tile_renders(product_lib, oracle, gpu_device):
    """A tile of 2^22 camera samples or more (512x512 at 16 spp here) is traced in
    ten pixel ranges and the trace updates whose sample indices a range completes
    go out as soon as that range is done (the reference reports while it traces,
    main.go:120-154).  The documents are the same as for a tile traced as a
    whole: progress values and running ray counts equal the oracle's, the image
    equals the oracle's tile."""
    events = []
    b = backend.Backend(device=gpu_device, progressUpdate=events.append)
    b.initialize(_context_json())
    b.loadBVH(b.buildBVH())
    sc = cornell()
    bvh = api.Bvh.build(product_mesh(sc), True, 6, 16)
    osc = oracle_scene(oracle, sc, bvh_export=bvh.export())
    osc.set_sample_tables(api.default_sample_table(12345, 0xC0FFEE01),
                          api.default_sample_table(100001, 0xC0FFEE02))
    cam = {"x": PRESET["x"], "y": PRESET["y"], "z": PRESET["z"]}
    d = scenes.make_pass_dict(512, 512, cam, spp=16, bounces=1, light_samples=1, gamma=PRESET["gamma"],
                              intensity=PRESET["lightIntensity"], seed=5, task_id=3)
    assert d["Width"] * d["Height"] * 16 >= 1 << 22
    doc = json.loads(b.render(json.dumps(d)))
    assert doc["exitCode"] == 0
    assert b.progress_deliveries >= 5, b.progress_deliveries        # not all at the end
    want = oracle.render_progress(osc, oracle.pass_from_dict(d))
    got = [json.loads(e) for e in events]
    trace = [e for e in got if e["event"] == "trace"]
    assert len(trace) == len(want) == 11
    for e, (progress, rays) in zip(trace, want):
        assert np.float32(e["progress"]) == np.float32(progress), (e, progress)
        assert e["rays"] == rays, (e, rays)
    assert b.rays == want[-1][1]
    ref = osc.render_tile(oracle.pass_from_dict(d))
    assert np.array_equal(backend.decode_image(doc), ref["rgba"])
    # a small tile still reports once, after it is done
    del events[:]
    small = _tasks(spp=2)[0]
    small["RenderKey"] = 9
    assert json.loads(b.render(json.dumps(small)))["exitCode"] == 0
    assert b.progress_deliveries == 1
