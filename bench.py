#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s (primary + shadow + bounce rays, every
rayCast of the reference counts, process/trace.go:122) for one tiled
path-traced frame of the synthetic 1M-triangle room.

    python bench.py --gpus N --steps K --warmup W

One step = one whole 3840x2160 frame at --spp samples per pixel, 4 diffuse
bounces, 1 light sample per vertex, cut into the frontend's 16x16 tile grid
(Renderer.js:388-405).  Tiles are dealt block-cyclically to the N ranks (one
process per GPU), every rank renders its tiles with the scene replicated in its
HBM, the RGBA8 tiles are gathered to rank 0 over RCCL and the ray counters are
all-reduced.  The frame is fixed, so scaling is strong.

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      the traversal kernel against the ceiling that bounds it: VALU
                lane-instruction issue while the scene is cache resident,
                measured fabric bytes against the HBM peak beyond that; launch
                times are measured live with HIP events on the launch stream,
                per-ray instruction and byte figures come from counter passes
                kept under profiles/ (the line says which, and whether the
                kernel sources have changed since)
  cpu_baseline  the CPU oracle (a scalar restatement of the reference
                algorithm) timed on rank 0's host cores over a bounded sample
                of the same frame (N=1 only)
  extra         two more legs timed after the headline region (N=1, default
                workload and mode only): the same frame in the bit-exact
                reference-order mode at 16 spp, and the 10M-triangle 8K
                configuration at 16 spp in the ordered mode
  ranks         (N>1) per-rank render and gather times and tile counts
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0
# VALU lane-instruction issue peak: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz
VALU_PEAK_TLANE = 256 * 4 * 16 * 2.4e9 / 1e12
INFINITY_CACHE_MB = 256.0
# files whose content decides what the traversal kernels do per ray: the counter
# passes under profiles/ carry the hash of these, and the line says when they changed
KERNEL_SOURCES = ["raytracer_amd/csrc/rt_kernels.hip", "raytracer_amd/csrc/rt_core.h",
                  "raytracer_amd/csrc/flatten.cpp", "raytracer_amd/build.py"]
PROFILE_ROUNDS = ["r03", "r02"]            # newest first

WORKLOADS = {
    # name: grid, triangles, width, height, tiles per dim, leaf, depth, intensity
    "cfg3": dict(grid=706, tris=1_000_000, width=3840, height=2160, tiles=16, leaf=8, depth=40,
                 bounces=4, light_samples=1, intensity=6.0,
                 label="1M-triangle room mesh (SAH BVH leaf 8), 3840x2160, 4 diffuse bounces, "
                       "1 light sample, 16x16 tiles"),
    # the HBM-resident configuration: 0.5 GB of triangles + 0.2 GB of tree records per GPU
    "cfg4": dict(grid=2236, tris=10_000_000, width=7680, height=4320, tiles=32, leaf=8, depth=48,
                 bounces=4, light_samples=1, intensity=6.0, seed=0x5EED0004,
                 label="10M-triangle room mesh (SAH BVH leaf 8), 7680x4320, 4 diffuse bounces, "
                       "1 light sample, 32x32 tiles"),
    "small": dict(grid=96, tris=20_000, width=640, height=360, tiles=4, leaf=8, depth=40,
                  bounces=4, light_samples=1, intensity=6.0,
                  label="20k-triangle room mesh, 640x360 (functional check only)"),
}
KERNEL_OF = {"fast": "k_trace_fast", "strict": "k_trace_ref<float>", "fp64": "k_trace_ref<double>"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel of one frame")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["strict", "fast", "fp64"],
                    help="traversal kernel: fast = ordered traversal (default), strict = the "
                         "reference's visit order (bit-exact mode), fp64 = validation mode")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="budget of the CPU-baseline sample (0 disables it)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, host cores)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the two extra legs (bit-exact mode on the same frame, 10M-triangle 8K frame)")
    ap.add_argument("--extra", default="strict_cfg3,fast_cfg4",
                    help="which extra legs a default run adds (comma separated)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (tiles staged through host memory, all ranks may "
                         "share one GPU) is for rehearsing the multi-rank path on a 1-GPU box")
    ap.add_argument("--dynamic", action="store_true",
                    help="ranks pull tiles from a shared counter instead of the static block-cyclic deal")
    ap.add_argument("--chunk", type=int, default=0,
                    help="tiles taken per pull with --dynamic (0 = enough for about 32 M camera samples per call)")
    ap.add_argument("--tune", default="", help="stack_entries,refill_below,blocks_per_cu (experiments)")
    ap.add_argument("--max-paths", type=int, default=1 << 29,
                    help="camera samples traced together (0 = the library's default of 64 M = 24 GB of path "
                         "workspace; the bench takes 512 M = 192 GB of the GPU's 288 GB, so that a 64-spp 4K "
                         "frame is ONE batch: every traversal launch ends with a drain phase, and one batch "
                         "instead of eight is 3.5 %% faster; a batch that does not fit free memory is halved "
                         "by the library, and the line reports the batch actually traced)")
    ap.add_argument("--pretend-world", type=int, default=0,
                    help="experiment on one GPU: render only the tiles rank 0 of this many ranks would own, "
                         "no collectives (estimates the per-rank time of a larger job)")
    return ap.parse_args()


def bytes_per_ray(nodes_visited, tris_fetched, rays, node_bytes=56.0):
    """Algorithmic traffic of one rayCast: 32 B ray in, 16 B hit out, 48 B per
    leaf triangle fetched, and per interior node whose children are tested 56 B
    in the reference's binary tree (two 24-B boxes + two links) or the whole
    64-B record of the ordered traversal's 4-wide tree."""
    return 32.0 + 16.0 + node_bytes * nodes_visited / rays + 48.0 * tris_fetched / rays


def kernel_source_sha():
    """Hash of the files that decide what the traversal kernels do per ray."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profile_entry(kind, workload, mode):
    """(entry, file name) of the newest counter pass under profiles/ for this
    workload and mode; kind is "valu_per_ray" or "traffic"."""
    for rnd in PROFILE_ROUNDS:
        name = f"{rnd}_{kind}.json"
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            entry = json.load(f).get(workload, {}).get(mode)
        if entry:
            return entry, name
    return None, None


def cpu_baseline(sc, bvh_export, tables, wl, args, tile_list):
    """Oracle render of a spread sample of tiles at 1 spp on host threads."""
    from oracle import oracle as O
    from raytracer_amd import scenes
    n_threads = args.cpu_threads or min(16, os.cpu_count() or 1)
    osc = O.Scene(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
    e = bvh_export
    osc.set_bvh(e["is_leaf"], e["start"], e["end"], e["bmin"], e["bmax"], e["order"])
    osc.set_sample_tables(*tables)
    # visit tiles in a stride that spreads over the frame
    order = [tile_list[(i * 37) % len(tile_list)] for i in range(len(tile_list))]
    lock = threading.Lock()

    def timed(n_workers, seconds):
        state = {"next": 0, "rays": 0, "tiles": 0}
        deadline = time.perf_counter() + seconds
        t0 = time.perf_counter()
        threads = [threading.Thread(target=worker, args=(state, deadline)) for _ in range(n_workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return state, time.perf_counter() - t0

    def worker(state, deadline):
        while True:
            with lock:
                if state["next"] >= len(order) or time.perf_counter() > deadline:
                    return
                t = order[state["next"]]
                state["next"] += 1
            tid, x, y, w, h = t
            d = scenes.make_pass_dict(wl["width"], wl["height"], sc["camera"], spp=1,
                                      bounces=wl["bounces"], light_samples=wl["light_samples"],
                                      intensity=wl["intensity"], seed=1, x_offset=x, y_offset=y,
                                      tile_w=w, tile_h=h, task_id=tid)
            out = osc.render_tile(O.pass_from_dict(d))
            with lock:
                state["rays"] += out["counters"]["rays"]
                state["tiles"] += 1

    state, dt = timed(n_threads, args.cpu_seconds * 0.4)
    # every core this process may run on, one thread per core (capped at the tile count)
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    n_all = max(1, min(usable, len(order)))
    every, dt_all = timed(n_all, args.cpu_seconds * 0.4)
    # the same tiles on ONE thread (the reference runs one wasm instance per worker), a few seconds
    one, dt1 = timed(1, min(4.0, args.cpu_seconds * 0.2))
    return {"value": state["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": n_threads,
            "kind": "port",
            "all_cores_value": round(every["rays"] / dt_all / 1e6, 3),
            "all_cores": n_all, "os_cpu_count": os.cpu_count(),
            "single_core_value": round(one["rays"] / dt1 / 1e6, 3) if one["rays"] else None,
            "go": go_baseline(),
            "sample": f"{state['tiles']} of {len(tile_list)} tiles of the same frame at 1 spp, "
                      f"{state['rays']} rays in {dt:.1f} s, scalar C restatement of the "
                      f"reference algorithm, one thread per tile worker ({n_threads} threads: the preset's 16 "
                      f"workers); all_cores_value: {every['tiles']} tiles, {every['rays']} rays in {dt_all:.1f} s "
                      f"on {n_all} threads (cores this process may use; the machine reports {os.cpu_count()}); "
                      f"single_core_value: {one['tiles']} tiles, {one['rays']} rays in {dt1:.1f} s on one thread"}


def go_baseline():
    """The stdlib-only Go restatement (go/cpu_restatement) keeps the reference's
    data layout -- pointer tree, []*Triangle, a heap-allocated ray per cast -- to
    price "Go on the CPU" honestly.  It needs a Go toolchain, which the build
    environment lacks; when `go version` works it is built and run on its own
    synthetic scene, otherwise the field says so."""
    import shutil
    import subprocess
    if shutil.which("go") is None:
        return "absent"
    src = os.path.join(ROOT, "go", "cpu_restatement")
    try:
        out = subprocess.run(["go", "run", ".", "-tris", "100000", "-rays", "2000000"], cwd=src,
                             capture_output=True, text=True, timeout=180)
        if out.returncode != 0:
            return {"error": out.stderr[-300:]}
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001 - reported in the line, never fatal for the bench
        return {"error": str(e)[:300]}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with
    torch.distributed.run as a CHILD process (the parent has not touched the GPU
    and never execs) and hand its exit code on.  Never falls back to one rank."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()          # does not initialise the GPU
    if args.backend == "nccl" and n_dev < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: only {n_dev} GPU(s) visible; refusing to run fewer ranks "
                         f"than asked for (use --backend gloo to rehearse several ranks on one GPU)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Job:
    """What every leg of one process shares: rank layout, collectives, device."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the ray-casting path has no CPU fallback")
        if args.backend == "gloo":
            self.local_rank = self.local_rank % torch.cuda.device_count()
        torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group("gloo")
        self.coll_dev = "cuda" if args.backend == "nccl" else "cpu"
        self.pretend = args.pretend_world if self.world == 1 else 0
        self.stream = torch.cuda.current_stream().cuda_stream

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


class Workload:
    """One synthetic scene resident on this rank's GPU plus the frame's tile list."""

    def __init__(self, job, name):
        from raytracer_amd import api, scenes
        self.job, self.name, self.wl = job, name, WORKLOADS[name]
        wl, dist, rank, world = self.wl, job.dist, job.rank, job.world
        t0 = time.perf_counter()
        self.sc = scenes.room_mesh(wl["grid"], seed=wl.get("seed", 0x5EED0002), total=wl["tris"])
        sc = self.sc
        self.mesh = api.Mesh(sc["verts"], sc["material"], sc["group_index"], sc["kd"], sc["is_light"])
        t_mesh = time.perf_counter()
        if world == 1:
            self.bvh = api.Bvh.build(self.mesh, True, wl["leaf"], wl["depth"])
        else:
            # the reference's flow: worker 0 builds the tree, every worker loads it
            # (main.go:65-90, Renderer.js:235-297); here through the binary cache file
            import tempfile
            cache = os.path.join(tempfile.gettempdir(),
                                 f"rt_bench_{os.environ.get('MASTER_PORT', '0')}_{name}.bvh")
            self.bvh = None
            if rank == 0:
                self.bvh = api.Bvh.build(self.mesh, True, wl["leaf"], wl["depth"])
                try:
                    self.bvh.save(cache)
                except Exception:                      # noqa: BLE001 - the others then build for themselves
                    pass
            dist.barrier()
            if rank != 0:
                try:
                    self.bvh = api.Bvh.load(self.mesh, cache)
                except Exception:                      # noqa: BLE001
                    self.bvh = api.Bvh.build(self.mesh, True, wl["leaf"], wl["depth"])
            dist.barrier()
            if rank == 0 and os.path.exists(cache):
                os.remove(cache)
        t_bvh = time.perf_counter()
        self.scene = api.Scene(self.mesh, self.bvh, job.local_rank)
        self.setup_s = time.perf_counter() - t0
        # where the set-up time goes: synthetic mesh, SAH build (or load of rank 0's tree), flatten + upload
        self.setup_parts = {"mesh_s": round(t_mesh - t0, 2), "bvh_s": round(t_bvh - t_mesh, 2),
                            "scene_s": round(time.perf_counter() - t_bvh, 2)}
        args = job.args
        if args.tune:
            self.scene.set_tuning(*[int(x) for x in args.tune.split(",")])
        if args.max_paths:
            self.scene.set_max_paths(args.max_paths)
        self.tables = (api.default_sample_table(12345, 0xC0FFEE01), api.default_sample_table(100001, 0xC0FFEE02))
        self.tile_list = scenes.tiles(wl["width"], wl["height"], wl["tiles"])

    def release(self):
        """Frees the scene's device memory (geometry and path workspace)."""
        self.scene = None
        import gc
        gc.collect()
        self.job.torch.cuda.empty_cache()


def run_leg(job, W, mode_name, spp, steps, warmup):
    """Times `steps` frames of workload W in one traversal mode and returns the
    measured figures plus the roofline object of the traversal kernel."""
    from raytracer_amd import api, scenes, tiles
    torch, dist, args = job.torch, job.dist, job.args
    rank, world, pretend, coll_dev, stream = job.rank, job.world, job.pretend, job.coll_dev, job.stream
    wl, sc, scene, tile_list = W.wl, W.sc, W.scene, W.tile_list
    mode = {"strict": api.MODE_STRICT_F32, "fast": api.MODE_FAST_F32, "fp64": api.MODE_FP64}[mode_name]

    my_tiles = [tile_list[i] for i in tiles.static_assignment(len(tile_list), pretend or world, rank)]
    # every rank's tile images live back to back in one buffer; ranks own equal
    # numbers of pixels only when the tile grid divides evenly, so pad to the maximum
    px_of = [sum(t[3] * t[4] for t in tile_list[r::(pretend or world)]) for r in range(world)]
    my_px, max_px = px_of[rank], max(px_of)
    if args.dynamic:
        my_px = max_px = sum(t[3] * t[4] for t in tile_list)   # a rank may end up with any tile
    d_accum = torch.empty((max(my_px, 1) * 3,), dtype=torch.float32, device="cuda")
    d_tiles = torch.zeros((max_px * 4,), dtype=torch.uint8, device="cuda")
    frame = torch.zeros((wl["height"], wl["width"], 4), dtype=torch.uint8, device="cuda")
    frame_host = torch.zeros((wl["height"], wl["width"], 4), dtype=torch.uint8) if coll_dev == "cpu" else None
    queue = None
    if args.dynamic and world > 1:
        chunk = args.chunk or tiles.pull_chunk(len(tile_list), world, tile_list[0][3] * tile_list[0][4] * spp)
        queue = tiles.TileQueue(dist.distributed_c10d._get_default_store(), len(tile_list), chunk=chunk)

    def make_pass(tile, n_spp, seed):
        tid, x, y, w, h = tile
        return api.pass_from_dict(scenes.make_pass_dict(
            wl["width"], wl["height"], sc["camera"], spp=n_spp, bounces=wl["bounces"],
            light_samples=wl["light_samples"], intensity=wl["intensity"], seed=seed,
            x_offset=x, y_offset=y, tile_w=w, tile_h=h, task_id=tid))

    pass_cache = {}

    def render_tiles(group, step, off_px=0):
        """One call for a whole group of tiles: one ray queue per path depth for
        the group, so small tiles still fill the GPU.  The pass records of a group
        are built once; a new frame only changes the per-tile seeds (the frontend
        sends base seed + task id, Renderer.js:423)."""
        key = group[0][0], len(group)   # (first tile id, count) names a group: pulls are contiguous id ranges
        arr = pass_cache.get(key)
        if arr is None:
            arr = pass_cache[key] = api.pass_array([make_pass(t, spp, 0) for t in group])
        for i, t in enumerate(group):
            arr[i].rng_seed = 1 + 1000 * step + t[0]
        return scene.render_tiles_device(arr, d_accum.data_ptr() + off_px * 12, d_tiles.data_ptr() + off_px * 4,
                                         mode=mode, stream=stream, want_stats=True)

    def run_step(step, collect):
        rays, kernel_ms, launches, n_pulls, n_tiles = 0, 0.0, 0, 0, 0
        t_start = time.perf_counter()
        if queue is None:
            st = render_tiles(my_tiles, step)       # returns after the ray-count read-back: the tiles are done
            rays, kernel_ms, launches, n_tiles = st["rays"], st["kernel_ms"], st["kernel_launches"], len(my_tiles)
            t_rendered = time.perf_counter()
            if not args.no_gather and not pretend:
                if coll_dev == "cuda":
                    tiles.gather_static(d_tiles, tile_list, world, rank, frame)
                else:
                    tiles.gather_static(d_tiles.cpu(), tile_list, world, rank, frame_host)
        else:
            # pull-based: take the next chunk of tiles until the list is empty; the images of
            # the chunks line up back to back in this rank's tile buffer
            queue.reset(epoch=step)
            my_ids, off_px = [], 0
            while True:
                got = queue.pull()
                if not got:
                    break
                n_pulls += 1
                group = [tile_list[i] for i in got]
                st = render_tiles(group, step, off_px)
                rays += st["rays"]
                kernel_ms += st["kernel_ms"]
                launches += st["kernel_launches"]
                my_ids += got
                off_px += sum(t[3] * t[4] for t in group)
            n_tiles = len(my_ids)
            t_rendered = time.perf_counter()
            if not args.no_gather:
                if coll_dev == "cuda":
                    tiles.gather_dynamic(d_tiles, my_ids, tile_list, world, rank, frame)
                else:
                    tiles.gather_dynamic(d_tiles.cpu(), my_ids, tile_list, world, rank, frame_host)
        if world > 1:
            torch.cuda.synchronize()                # the exchange is asynchronous on the stream: time it here
        t_end = time.perf_counter()
        if collect is not None:
            collect["rays"] += rays
            collect["kernel_ms"] += kernel_ms
            collect["launches"] += launches
            collect["render_s"] += t_rendered - t_start
            collect["gather_s"] += t_end - t_rendered
            collect["pulls"] += n_pulls
            collect["tiles"] += n_tiles
        return rays

    # ---- bytes/ray of this workload from the counting kernel on a sample tile (untimed)
    scene.set_counters(True)
    sample_tile = tile_list[(len(tile_list) // 2) + wl["tiles"] // 2]
    st = scene.render_tile_device(make_pass(sample_tile, 1, 1), d_accum.data_ptr(),
                                  d_tiles.data_ptr(), mode=mode, stream=stream, want_stats=True)
    # the same sample tile in the reference's visit order: the byte count a literal
    # left-first WalkNode (models/bvh.go:117-144) needs for these rays
    st_ref = st if mode == api.MODE_STRICT_F32 else scene.render_tile_device(
        make_pass(sample_tile, 1, 1), d_accum.data_ptr(), d_tiles.data_ptr(),
        mode=api.MODE_STRICT_F32, stream=stream, want_stats=True)
    scene.set_counters(False)
    bpr = bytes_per_ray(st["nodes_visited"], st["tris_fetched"], max(st["rays"], 1),
                        64.0 if mode == api.MODE_FAST_F32 else 56.0)
    bpr_ref = bytes_per_ray(st_ref["nodes_visited"], st_ref["tris_fetched"], max(st_ref["rays"], 1))
    nodes_per_ray = st["nodes_visited"] / max(st["rays"], 1)
    tris_per_ray = st["tris_fetched"] / max(st["rays"], 1)

    for w_ in range(warmup):
        run_step(-1 - w_, None)
    acc = {"rays": 0, "kernel_ms": 0.0, "launches": 0, "render_s": 0.0, "gather_s": 0.0, "pulls": 0, "tiles": 0}
    job.barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        run_step(s, acc)
    job.barrier()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([float(acc["rays"]), elapsed, acc["kernel_ms"]], dtype=torch.float64, device=coll_dev)
    per_rank = None
    if world > 1:
        t_max = tot.clone()
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(t_max[1])
        kernel_ms = float(t_max[2])
        total_rays = float(tot[0])
        # what each rank spent where, so that a shortfall at N ranks can be attributed
        mine = torch.tensor([acc["render_s"] * 1e3 / steps, acc["gather_s"] * 1e3 / steps, acc["kernel_ms"] / steps,
                             acc["tiles"] / steps, acc["pulls"] / steps, acc["rays"] / steps],
                            dtype=torch.float64, device=coll_dev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        cols = list(zip(*[e.tolist() for e in every]))
        per_rank = {"render_ms_per_step": [round(v, 3) for v in cols[0]],
                    "gather_ms_per_step": [round(v, 3) for v in cols[1]],
                    "traversal_ms_per_step": [round(v, 3) for v in cols[2]],
                    "tiles_per_step": [round(v, 2) for v in cols[3]],
                    "pulls_per_step": [round(v, 2) for v in cols[4]],
                    "rays_per_step": [round(v, 1) for v in cols[5]],
                    "render_ms_min_max": [round(min(cols[0]), 3), round(max(cols[0]), 3)],
                    "gather_ms_min_max": [round(min(cols[1]), 3), round(max(cols[1]), 3)],
                    "note": "render = host clock around this rank's render calls (they return after the ray-count "
                            "read-back); gather = the exchange step up to a device synchronise, which on every rank "
                            "but the slowest includes waiting for the slowest"}
    else:
        total_rays, kernel_ms = float(acc["rays"]), acc["kernel_ms"]

    if rank != 0:
        return None
    info = scene.info()
    # what the library holds in HBM for the scene: triangle records, both tree views, shading words, tables
    scene_mb = info["total_device_bytes"] / 1e6
    value = total_rays / elapsed / 1e6
    # roofline of the traversal kernel, measured on this rank: rays over the time its
    # traversal launches took (HIP events on the launch stream, inside the timed region)
    rank_rays = float(acc["rays"])
    trav_s = acc["kernel_ms"] * 1e-3
    rays_per_s = rank_rays / trav_s if trav_s > 0 else 0.0
    algorithmic = rays_per_s * bpr / 1e9                       # GB/s of counted record + triangle fetches

    # fabric bytes per ray and VALU lane-instructions per ray are counter measurements of this
    # kernel on this workload, taken in separate rocprofv3 --pmc runs and kept under profiles/
    fabric, fabric_file = profile_entry("traffic", W.name, mode_name)
    valu, valu_file = profile_entry("valu_per_ray", W.name, mode_name)
    sha_now = kernel_source_sha()
    head = {}
    for label, entry, fname in (("valu", valu, valu_file), ("traffic", fabric, fabric_file)):
        if entry:
            head[label] = {"file": f"profiles/{fname}", "git_head": entry.get("git_head"),
                           "rt_version": entry.get("rt_version"), "kernel_source_sha": entry.get("kernel_source_sha"),
                           "matches_this_source": entry.get("kernel_source_sha") == sha_now}
    stale = [k for k, v in head.items() if not v["matches_this_source"]]
    traffic = traffic_src = None
    if fabric and acc["launches"]:
        traffic = round(fabric["hbm_bytes_per_ray"] * rank_rays / acc["launches"], 1)
        traffic_src = (f"profiles/{fabric_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per ray) "
                       "x this run's rays per launch")
    hbm_rate = fabric["hbm_bytes_per_ray"] * rays_per_s / 1e9 if fabric else None
    resident = scene_mb < INFINITY_CACHE_MB
    roof = {"kernel": KERNEL_OF[mode_name], "traffic": traffic, "traffic_source": traffic_src,
            "rays_per_s_in_kernel": round(rays_per_s, 1),
            "bytes_per_ray": round(bpr, 1), "bytes_per_ray_reference_order": round(bpr_ref, 1),
            "algorithmic_GBps": round(algorithmic, 1),
            "algorithmic_GBps_reference_order": round(algorithmic * bpr_ref / bpr, 1) if bpr > 0 else 0.0,
            "nodes_per_ray": round(nodes_per_ray, 2), "tris_per_ray": round(tris_per_ray, 2),
            "kernel_ms_per_step": round(kernel_ms / steps, 3),
            "kernel_launches_per_step": acc["launches"] // max(steps, 1),
            "kernel_avg_launch_ms": round(acc["kernel_ms"] / max(acc["launches"], 1), 4),
            "scene_mb": round(scene_mb, 1),
            "profile_head": {"kernel_source_sha_now": sha_now, "rt_version_now": api.version(), **head},
            "profile_warning": (f"the kernel sources changed after the counter pass of: {', '.join(stale)} -- "
                                "per-ray constants below are the older kernel's") if stale else None}
    if resident:
        # the scene lives in L2 / Infinity Cache: HBM is not the ceiling.  The kernel is bound by
        # VALU issue (profiles/): achieved = active lane-instructions per ray (SQ_THREAD_CYCLES_VALU
        # / rays) x rays/s against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
        lane_per_ray = valu["lane_inst_per_ray"] if valu else None
        ach = lane_per_ray * rays_per_s / 1e12 if lane_per_ray else None
        roof.update({"bound": "valu", "unit": "Tlane-inst/s", "peak": round(VALU_PEAK_TLANE, 2),
                     "achieved": round(ach, 3) if ach else None,
                     "frac": round(ach / VALU_PEAK_TLANE, 4) if ach else None,
                     "lane_inst_per_ray": lane_per_ray,
                     "wave_inst_per_ray": valu.get("wave_inst_per_ray") if valu else None,
                     "avg_active_lanes": valu.get("avg_active_lanes") if valu else None,
                     "lane_inst_source": f"profiles/{valu_file} (tools/pmc_to_profiles.py valu)" if valu else None})
        roof["hbm"] = {"achieved": round(hbm_rate, 1) if hbm_rate else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": round(hbm_rate / HBM_PEAK_GBPS, 4) if hbm_rate else None,
                       "note": "measured fabric bytes, not algorithmic ones"}
    else:
        # beyond the Infinity Cache: HBM, priced at the MEASURED fabric bytes
        roof.update({"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                     "achieved": round(hbm_rate, 1) if hbm_rate else None,
                     "frac": round(hbm_rate / HBM_PEAK_GBPS, 4) if hbm_rate else None})
    # the batch the library actually traced: the bound it holds now (lowered for good when a
    # batch did not fit device memory), capped by the camera samples one call had
    asked = int(args.max_paths) if args.max_paths else 1 << 26
    samples_per_call = max(sum(t[3] * t[4] for t in my_tiles), 1) * spp
    effective = min(int(info["max_paths_in_flight"]), samples_per_call)
    return {"value": round(value, 3), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup,
            "spp": spp, "mode": mode_name, "rays_per_step": total_rays / steps, "total_rays": total_rays,
            "queue": queue is not None, "resident": resident,
            "paths_per_batch": effective, "paths_per_batch_asked": asked,
            "batch_halved_by_library": int(info["max_paths_in_flight"]) < asked,
            "workspace_gb": round(info["workspace_bytes"] / 1e9, 2),
            "roofline": roof, "per_rank": per_rank,
            "nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray, "rays_per_s_in_kernel": rays_per_s}


def gather_ceiling(job, W, leg):
    """Beside the VALU roofline of a cache-resident scene: the gather rate of the
    cache hierarchy for tables of this scene's size (rt_gather_probe, untimed)."""
    from raytracer_amd import api
    info = W.scene.info()
    recs = {}
    for label, nbytes in (("ordered_view", info["ordered_bytes"]),
                          ("ordered_view_plus_triangles", info["ordered_bytes"] + info["tri_bytes"])):
        if nbytes:
            recs[label] = {"table_mb": round(nbytes / 1e6, 1),
                           "records_per_s": round(api.gather_probe(job.local_rank, nbytes), 1)}
    kernel_recs = (leg["nodes_per_ray"] + 0.75 * leg["tris_per_ray"]) * leg["rays_per_s_in_kernel"]   # a triangle record is 48 of 64 bytes
    uniform = recs.get("ordered_view_plus_triangles", {}).get("records_per_s")
    return {"uniform_gather": recs, "kernel_records_per_s": round(kernel_recs, 1),
            "kernel_over_uniform_gather": round(kernel_recs / uniform, 3) if uniform else None,
            "note": "rt_gather_probe: every lane chases UNIFORMLY random 64-byte records through a table "
                    "of the given size (measured by this run outside its timed region). The traversal's "
                    "fetches are not uniform -- the top of the tree is re-read by every ray and stays in "
                    "L1/L2 -- so its record rate can exceed the uniform figure; the ratio says how much "
                    "of its fetch stream the nearer cache levels absorb, it is not a fraction of a ceiling"}


def random_gather_ceiling(job, W, leg):
    """Beside the HBM roofline of a scene beyond the Infinity Cache: what the memory
    system delivers to the traversal's access pattern.  rt_gather_probe chases
    uniformly random 64-byte records through a table of this scene's size (ordered
    view + triangles); at 64 bytes per request that rate is far below the streaming
    peak, and it is the ceiling the kernel's fabric fetches stand against."""
    from raytracer_amd import api
    info = W.scene.info()
    nbytes = info["ordered_bytes"] + info["tri_bytes"]
    if not nbytes:
        return None
    recs = api.gather_probe(job.local_rank, nbytes)
    gbps = recs * 64.0 / 1e9
    fabric = leg["roofline"].get("achieved") if leg["roofline"].get("bound") == "hbm" else None
    return {"table_mb": round(nbytes / 1e6, 1), "records_per_s": round(recs, 1), "GBps": round(gbps, 1),
            "kernel_fabric_GBps": fabric,
            "kernel_fabric_over_gather": round(fabric / gbps, 3) if fabric and gbps else None,
            "note": "rt_gather_probe: every lane chases UNIFORMLY random 64-byte records through a table of "
                    "the scene's size (measured by this run outside its timed region); the kernel's fabric "
                    "bytes are the counter measurement of the roofline object (L2 hits do not count), so the "
                    "ratio says how close its misses run to what the memory system gives scattered 64-byte reads"}


def extra_object(W, leg):
    """One extra leg as it appears under "extra" in the line."""
    r = leg["roofline"]
    return {"workload": f"{W.name}: {W.wl['label']}, {leg['spp']} spp", "traversal_mode": leg["mode"],
            "value": leg["value"], "unit": "Mrays/s", "ms_per_step": leg["ms_per_step"], "steps": leg["steps"],
            "warmup": leg["warmup"], "rays_per_step": leg["rays_per_step"],
            "triangles": int(W.mesh.n_tris), "width": W.wl["width"], "height": W.wl["height"], "spp": leg["spp"],
            "paths_per_batch": leg["paths_per_batch"], "batch_halved_by_library": leg["batch_halved_by_library"],
            "workspace_gb": leg["workspace_gb"], "setup_s": round(W.setup_s, 2), "setup_parts": W.setup_parts,
            "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                               "rays_per_s_in_kernel", "kernel_ms_per_step",
                                               "kernel_launches_per_step", "kernel_avg_launch_ms", "nodes_per_ray",
                                               "tris_per_ray", "bytes_per_ray", "scene_mb", "lane_inst_per_ray",
                                               "hbm_random_gather", "profile_head", "profile_warning")
                         if k in r or k != "hbm_random_gather"}}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    job = Job(args)
    world, rank = job.world, job.rank

    W = Workload(job, args.workload)
    leg = run_leg(job, W, args.mode, args.spp, args.steps, args.warmup)

    out = None
    if rank == 0:
        wl, roof = W.wl, leg["roofline"]
        if leg["resident"]:
            roof["l2_mall"] = gather_ceiling(job, W, leg)
        else:
            roof["hbm_random_gather"] = random_gather_ceiling(job, W, leg)
        out = {
            "metric": "Mrays/sec (primary+secondary)",
            "value": leg["value"], "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": leg["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.mode != "fp64" else "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['label']}, {args.spp} spp",
                       "triangles": int(W.mesh.n_tris), "bvh_nodes": int(W.bvh.n_nodes),
                       "width": wl["width"], "height": wl["height"], "spp": args.spp,
                       "bounces": wl["bounces"], "light_samples": wl["light_samples"],
                       "tiles": len(W.tile_list),
                       "tile_assignment": "dynamic pull queue" if leg["queue"] else "block-cyclic",
                       "traversal_mode": args.mode, "parallelism": f"tiles/{world}gpu",
                       "paths_per_batch": leg["paths_per_batch"],
                       "paths_per_batch_asked": leg["paths_per_batch_asked"],
                       "batch_halved_by_library": leg["batch_halved_by_library"],
                       "workspace_gb": leg["workspace_gb"],
                       "rays_per_step": leg["rays_per_step"], "setup_s": round(W.setup_s, 2),
                       "setup_parts": W.setup_parts},
            "roofline": roof,
        }
        if leg["per_rank"]:
            out["ranks"] = leg["per_rank"]

    # ---- extra legs: the bit-exact mode on the same frame, and the 10M-triangle 8K frame
    want_extra = (world == 1 and not args.no_extra and not job.pretend and args.workload == "cfg3"
                  and args.mode == "fast" and not args.tune)
    extra = {}
    t_extra = time.perf_counter()
    if want_extra and "strict_cfg3" in args.extra:
        extra["strict_cfg3"] = extra_object(W, run_leg(job, W, "strict", 16, 2, 1))
    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(W.sc, W.bvh.export(), W.tables, W.wl, args, W.tile_list)
    if want_extra and "fast_cfg4" in args.extra:
        W.release()                                  # its 192 GB of path workspace make room for the next scene's
        W = None
        W4 = Workload(job, "cfg4")
        leg4 = run_leg(job, W4, "fast", 16, 1, 1)
        if not leg4["resident"]:
            leg4["roofline"]["hbm_random_gather"] = random_gather_ceiling(job, W4, leg4)
        extra["fast_cfg4"] = extra_object(W4, leg4)
        W4.release()
    if rank == 0:
        if extra:
            extra["seconds"] = round(time.perf_counter() - t_extra, 1)
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if world > 1:
        job.dist.barrier()
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
