"""Per-ray traversal counters of both kernels on the benchmark frame (1 spp)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes
sc = scenes.room_mesh(706, seed=0x5EED0002, total=1_000_000)
mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
bvh = api.Bvh.build(mesh, True, 8, 40)
scene = api.Scene(mesh, bvh, 0)
tl = scenes.tiles(3840, 2160, 16)
px = sum(t[3]*t[4] for t in tl)
acc = torch.empty((px*3,), dtype=torch.float32, device="cuda"); rgba = torch.zeros((px*4,), dtype=torch.uint8, device="cuda")
passes=[api.pass_from_dict(scenes.make_pass_dict(3840,2160,sc["camera"],spp=1,bounces=4,light_samples=1,intensity=6.0,seed=1,x_offset=x,y_offset=y,tile_w=w,tile_h=h,task_id=tid)) for (tid,x,y,w,h) in tl]
scene.set_counters(True)
for mode in (api.MODE_STRICT_F32, api.MODE_FAST_F32):
    st = scene.render_tiles_device(passes, acc.data_ptr(), rgba.data_ptr(), mode=mode, stream=torch.cuda.current_stream().cuda_stream, want_stats=True)
    r=st["rays"]; print(mode, {k: round(st[k]/r,2) for k in ("nodes_visited","box_tests","tris_fetched","tris_tested")})
