#!/usr/bin/env python3
"""The reference's demo workload -- the bundled Cornell box preset (reference
src/frontend/app/public/presets/cornell-box.json:4-31: 250x250, 25 samples per
pixel, 3 bounces, 6 light samples, 16 tiles, gamma 2.2, intensity 100) --
rendered twice:

  * tile by tile through render(passJSON) of the GPU backend (main.go:93-188
    as rebuilt: raytracer_amd.backend.Backend.render) in the bit-exact mode, and
  * by the CPU oracle's restatement of main.go:93-188, on 1 and on 16 threads
    (the preset's 16 workers),

asserts that the two frames are equal byte for byte and that the ray counters
agree, writes both images and the timings under profiles/:

    python tools/cornell_preset.py [OUT_DIR]       (needs a GPU; a few seconds; OUT_DIR defaults to profiles/)
      -> OUT_DIR/r03_cornell_preset_gpu.png, OUT_DIR/r03_cornell_preset_oracle.png,
         OUT_DIR/r03_cornell_preset.json

The ordered mode is timed beside the bit-exact one; its frame is compared with
the bit-exact frame and the differing pixels are counted.
"""
import json
import os
import struct
import sys
import threading
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PRESET = dict(width=250, height=250, x=-0.225, y=2.55, z=6.0, bounces=3, lightSampleRays=6, raysPerPixel=25,
              taskCount=16, workerCount=16, gamma=2.2, maxLeafSize=6, maxDepth=16, lightIntensity=100.0)


def write_png(path, rgba):
    """Minimal RGBA8 PNG writer (stored in one zlib stream, filter 0 per row)."""
    h, w, _ = rgba.shape
    raw = b"".join(b"\0" + rgba[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def tasks():
    from raytracer_amd import scenes
    n = int(np.sqrt(PRESET["taskCount"]))
    cam = {"x": PRESET["x"], "y": PRESET["y"], "z": PRESET["z"]}
    return [scenes.make_pass_dict(PRESET["width"], PRESET["height"], cam, spp=PRESET["raysPerPixel"],
                                  bounces=PRESET["bounces"], light_samples=PRESET["lightSampleRays"],
                                  gamma=PRESET["gamma"], intensity=PRESET["lightIntensity"], seed=1,
                                  x_offset=x, y_offset=y, tile_w=w, tile_h=h, task_id=tid)
            for tid, x, y, w, h in scenes.tiles(PRESET["width"], PRESET["height"], n)]


def gpu_frame(b, mode, key):
    """One frame: 16 render(passJSON) calls, JSON in, JSON + base64 out."""
    from raytracer_amd import backend
    b.set_mode(mode)
    frame = np.zeros((PRESET["height"], PRESET["width"], 4), np.uint8)
    rays0 = None
    t0 = time.perf_counter()
    for d in tasks():
        d["RenderKey"] = key                       # a new key resets the ray counter (main.go:107-110)
        doc = json.loads(b.render(json.dumps(d)))
        if doc["exitCode"] != 0:
            raise SystemExit(f"render failed: {doc['message']}")
        if rays0 is None:
            rays0 = 0
        frame[d["YOffset"]:d["YOffset"] + d["Height"], d["XOffset"]:d["XOffset"] + d["Width"]] = \
            backend.decode_image(doc)
    return frame, b.rays, time.perf_counter() - t0


def oracle_frame(osc, O, n_threads):
    frame = np.zeros((PRESET["height"], PRESET["width"], 4), np.uint8)
    todo = tasks()
    lock = threading.Lock()
    state = {"next": 0, "rays": 0}

    def worker():
        while True:
            with lock:
                if state["next"] >= len(todo):
                    return
                d = todo[state["next"]]
                state["next"] += 1
            out = osc.render_tile(O.pass_from_dict(d))
            with lock:
                state["rays"] += out["counters"]["rays"]
                frame[d["YOffset"]:d["YOffset"] + d["Height"], d["XOffset"]:d["XOffset"] + d["Width"]] = out["rgba"]

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker) for _ in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return frame, state["rays"], time.perf_counter() - t0


def main():
    from oracle import oracle as O
    from raytracer_amd import api, backend
    from util import GOLDEN, cornell, oracle_scene, product_mesh
    d = os.path.join(GOLDEN, "cornell-box")
    obj = open(os.path.join(d, "cornell-box.obj")).read()
    mtl = open(os.path.join(d, "cornell-box.mtl")).read()
    b = backend.Backend(device=0)
    b.initialize(json.dumps({"Debug": False, "UseBVH": True, "BVHMaxLeafSize": PRESET["maxLeafSize"],
                             "BVHMaxDepth": PRESET["maxDepth"], "Scene": {}, "ObjBuffer": obj, "MtlBuffer": mtl,
                             "RawTextures": [], "WorkerId": 0}))
    b.loadBVH(b.buildBVH())

    sc = cornell()
    bvh = api.Bvh.build(product_mesh(sc), True, PRESET["maxLeafSize"], PRESET["maxDepth"])
    osc = oracle_scene(O, sc, bvh_export=bvh.export())
    osc.set_sample_tables(api.default_sample_table(12345, 0xC0FFEE01), api.default_sample_table(100001, 0xC0FFEE02))

    key = 100
    gpu_frame(b, api.MODE_STRICT_F32, key)                       # warm-up (workspace allocation)
    strict = [gpu_frame(b, api.MODE_STRICT_F32, key + 1 + i) for i in range(5)]
    fast = [gpu_frame(b, api.MODE_FAST_F32, key + 10 + i) for i in range(5)]
    img, rays, _ = strict[0]
    o16_img, o16_rays, o16_s = oracle_frame(osc, O, PRESET["workerCount"])
    o1_img, o1_rays, o1_s = oracle_frame(osc, O, 1)

    assert np.array_equal(o1_img, o16_img), "the oracle's frame depends on the thread count"
    assert np.array_equal(img, o16_img), "GPU frame differs from the oracle's"
    assert rays == o16_rays == o1_rays, (rays, o16_rays, o1_rays)
    assert all(np.array_equal(f, img) and r == rays for f, r, _ in strict), "the GPU frame is not reproducible"
    fast_img, fast_rays, _ = fast[0]
    differing = int((fast_img != img).any(axis=2).sum())

    P = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
    os.makedirs(P, exist_ok=True)
    write_png(os.path.join(P, "r03_cornell_preset_gpu.png"), img)
    write_png(os.path.join(P, "r03_cornell_preset_oracle.png"), o16_img)
    best = lambda runs: min(t for _, _, t in runs)             # noqa: E731
    doc = {
        "workload": "cfg0: bundled Cornell box (34 triangles), preset cornell-box.json: 250x250, 25 spp, 3 bounces, "
                    "6 light samples, 16 tiles of 62/64 px, gamma 2.2, intensity 100, RNGSeed = 1 + TaskID",
        "rays_per_frame": int(rays),
        "ui_estimate_rays": 250 * 250 * 25 * (1 + 6 + 6 * 3),
        "frames_equal_byte_for_byte": True, "ray_counters_equal": True,
        "gpu_bit_exact_mode": {"ms_per_frame": round(best(strict) * 1e3, 2), "Mrays_per_s": round(rays / best(strict) / 1e6, 1),
                               "how": "16 render(passJSON) calls per frame through the backend entry point, JSON in, "
                                      "JSON + base64 image out, best of 5 frames"},
        "gpu_ordered_mode": {"ms_per_frame": round(best(fast) * 1e3, 2), "Mrays_per_s": round(fast_rays / best(fast) / 1e6, 1),
                             "rays_per_frame": int(fast_rays), "pixels_differing_from_bit_exact_frame": differing},
        "cpu_oracle_16_threads": {"s_per_frame": round(o16_s, 3), "Mrays_per_s": round(o16_rays / o16_s / 1e6, 3)},
        "cpu_oracle_1_thread": {"s_per_frame": round(o1_s, 3), "Mrays_per_s": round(o1_rays / o1_s / 1e6, 3)},
        "host_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
        "rt_version": api.version(), "git_head": os.environ.get("RT_GIT_HEAD") or None,
        "images": ["profiles/r03_cornell_preset_gpu.png", "profiles/r03_cornell_preset_oracle.png"],
    }
    with open(os.path.join(P, "r03_cornell_preset.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
