import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from raytracer_amd import api, scenes
from util import random_rays
sc = scenes.soup(2000, seed=12)
mesh = api.Mesh(sc["verts"]); bvh = api.Bvh.build(mesh, True, 6, 24)
base = api.Scene(mesh, bvh, 0)
for n in (64, 300, 5000, 100000):
    rays = random_rays(n, seed=3)
    want, st0 = base.trace_rays(rays, mode=api.MODE_FAST_F32, counters=True)
    os.environ["RT_POOL"] = "1,24,48,16"
    scene = api.Scene(mesh, bvh, 0)
    os.environ.pop("RT_POOL")
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8).copy()).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    d_cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    scene.trace_rays_device(d_rays.data_ptr(), n, d_hits.data_ptr(), mode=api.MODE_FAST_F32,
                            stream=torch.cuda.current_stream().cuda_stream, d_counters_ptr=d_cnt.data_ptr())
    torch.cuda.synchronize()
    got = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
    c = d_cnt.cpu().numpy()
    print(n, "same", np.array_equal(got.view(np.uint32), want.view(np.uint32)), "mismatch", int((got["tri_pos"] != want["tri_pos"]).sum()),
          "counters", c.tolist(), [hex(int(x)) for x in c[5:8]], "base", st0["nodes_visited"], st0["tris_fetched"], flush=True)
