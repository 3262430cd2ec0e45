"""Latency of the reference-style call pattern: one render() per tile of the
Cornell preset (250x250, 4x4 tiles, 25 spp, 6 light samples, 3 bounces),
JSON in, JSON + base64 out."""
import sys, time, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from raytracer_amd import backend, scenes, api
from util import GOLDEN
import os
d = os.path.join(GOLDEN, "cornell-box")
obj = open(os.path.join(d, "cornell-box.obj")).read(); mtl = open(os.path.join(d, "cornell-box.mtl")).read()
b = backend.Backend(device=0)
b.initialize(json.dumps({"UseBVH": True, "BVHMaxLeafSize": 6, "BVHMaxDepth": 16, "ObjBuffer": obj, "MtlBuffer": mtl}))
b.loadBVH(b.buildBVH())
cam = {"x": -0.225, "y": 2.55, "z": 6.0}
tl = scenes.tiles(250, 250, 4)
for mode in (api.MODE_STRICT_F32, api.MODE_FAST_F32):
    b.set_mode(mode)
    for rep in range(3):
        t0 = time.perf_counter(); rays0 = b.rays
        for (tid, x, y, w, h) in tl:
            p = scenes.make_pass_dict(250, 250, cam, spp=25, bounces=3, light_samples=6, intensity=100.0, seed=1, x_offset=x, y_offset=y, tile_w=w, tile_h=h, task_id=tid)
            p["RenderKey"] = 7
            doc = b.render(json.dumps(p))
        dt = time.perf_counter() - t0
        print("mode", mode, "frame of 16 tiles: %.1f ms, %.2f ms/tile, %.1f Mrays/s" % (dt*1e3, dt*1e3/16, (b.rays-rays0)/dt/1e6))
