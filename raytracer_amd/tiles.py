"""Tile sharding of one frame over the ranks of a node and the exchange step
that brings the finished tiles to rank 0.

The reference distributes a frame the same way over browser workers: the scene
and BVH are copied to every worker (frontend Renderer.js:213-228,281-297), the
frame is cut into a grid of tiles (Renderer.js:388-405), every finished worker
takes the next tile from a shared list (Renderer.js:565-568) and the page
composites the returned tiles by offset (RendererFrame.js:79-105).  Here a rank
is one process with one GPU; the tile list is dealt either statically
(block-cyclic) or through a shared counter that ranks pull from, and the tiles
come home through one collective (RCCL on GPUs, gloo on CPU tensors in tests).
"""
import torch
import torch.distributed as dist


def static_assignment(n_tiles, world, rank):
    """Block-cyclic: tile i belongs to rank i % world."""
    return list(range(rank, n_tiles, world))


class TileQueue:
    """Pull-based tile list shared by all ranks: the counterpart of the
    frontend's `renderTasks.pop()`.  Backed by the process group's key-value
    store (an atomic add on the store's host), so a rank that finishes early
    simply takes more tiles.  `chunk` tiles are taken per pull."""

    def __init__(self, store, n_tiles, chunk=1, key="rt_tile_cursor"):
        self.store, self.n_tiles, self.chunk, self.key = store, n_tiles, max(1, int(chunk)), key

    def reset(self, epoch):
        """Every rank calls this with the same epoch before a frame."""
        self.key_epoch = f"{self.key}_{epoch}"

    def pull(self):
        """Next chunk of tile indices, [] when the list is exhausted."""
        end = self.store.add(self.key_epoch, self.chunk)
        start = end - self.chunk
        return list(range(start, min(end, self.n_tiles))) if start < self.n_tiles else []


def tile_pixel_offsets(tiles):
    """Start pixel of every tile when a rank's tiles are stored back to back."""
    offs, total = [], 0
    for (_tid, _x, _y, w, h) in tiles:
        offs.append(total)
        total += w * h
    return offs, total


def _paste_uniform(frame, buf, mine):
    """All of a rank's tiles in ONE strided copy when they share one size and sit
    on a regular lattice of the frame (the common case: the frame divides evenly
    into the tile grid and ranks take every world-th tile).  Returns False when
    the tiles do not fit that pattern."""
    if not mine:
        return True
    _t, x0, y0, w, h = mine[0]
    if any(t[3] != w or t[4] != h for t in mine) or frame.shape[1] % w or frame.shape[0] % h:
        return False
    if any(t[1] % w or t[2] % h for t in mine):
        return False
    cols, rows = frame.shape[1] // w, frame.shape[0] // h
    grid = frame.view(rows, h, cols, w, 4).permute(0, 2, 1, 3, 4)      # (row, col, h, w, 4) view
    row = torch.tensor([t[2] // h for t in mine], device=frame.device)
    col = torch.tensor([t[1] // w for t in mine], device=frame.device)
    grid[row, col] = buf[:len(mine) * w * h * 4].view(len(mine), h, w, 4)
    return True


def gather_static(local_rgba, tile_list, world, rank, frame=None, group=None, force_collective=False):
    """Static assignment: every rank holds the RGBA8 images of its block-cyclic
    tiles back to back in `local_rgba` (uint8, padded to the largest per-rank
    size).  One gather to rank 0, which pastes every tile at its offset into
    `frame` ((H, W, 4) uint8).  Returns the frame on rank 0, None elsewhere.
    `force_collective` runs the collective even in a group of one rank (tests
    of the RCCL branch on a single GPU)."""
    if world > 1 or force_collective:
        bufs = [torch.empty_like(local_rgba) for _ in range(world)] if rank == 0 else None
        dist.gather(local_rgba, bufs, dst=0, group=group)
    else:
        bufs = [local_rgba]
    if rank != 0:
        return None
    for r, buf in enumerate(bufs):
        paste_tiles(frame, buf, tile_list[r::world])
    return frame


def paste_tiles(frame, buf, tiles):
    """Tiles stored back to back in `buf` (uint8 RGBA) go to their offsets in
    `frame` ((H, W, 4) uint8): one strided copy when they fit a lattice, a copy
    per tile otherwise."""
    if _paste_uniform(frame, buf, tiles):
        return
    off = 0
    for (_tid, x, y, w, h) in tiles:
        frame[y:y + h, x:x + w] = buf[off:off + w * h * 4].view(h, w, 4)
        off += w * h * 4


def gather_dynamic(local_rgba, my_ids, tile_list, world, rank, frame, group=None, force_collective=False):
    """Dynamic assignment: a rank holds the RGBA8 images of the tiles it happened
    to pull, back to back in pull order, and the list of their ids.  One small
    all-gather tells every rank who rendered what; then each rank sends exactly
    its own pixels to rank 0 (point to point, no padding to the worst case and
    no full-frame reduction), which pastes them.  All receives of rank 0 and
    every rank's send go out as ONE batch (`dist.batch_isend_irecv`: one RCCL
    group), so that the peers' transfers run side by side, each over its own
    xGMI link, instead of one after another.  Returns the frame on rank 0.
    `force_collective` runs the all-gather and a send-to-self / receive-from-self
    pair even in a group of one rank (tests of the RCCL branch on a single GPU)."""
    n_tiles = len(tile_list)
    dev = local_rgba.device
    ids = torch.full((n_tiles,), -1, dtype=torch.int32, device=dev)
    if my_ids:
        ids[:len(my_ids)] = torch.tensor(my_ids, dtype=torch.int32, device=dev)
    if world > 1 or force_collective:
        all_ids = [torch.empty_like(ids) for _ in range(world)]
        dist.all_gather(all_ids, ids, group=group)
    else:
        all_ids = [ids]
    my_bytes = sum(tile_list[i][3] * tile_list[i][4] for i in my_ids) * 4
    self_loop = force_collective and rank == 0       # rank 0 also sends to itself instead of copying
    ops, bufs, owned = [], [], None
    if (rank != 0 or self_loop) and my_bytes:
        ops.append(dist.P2POp(dist.isend, local_rgba[:my_bytes], 0, group))
    if rank == 0:
        owned = [[int(i) for i in t.tolist() if i >= 0] for t in all_ids]
        seen = sorted(i for o in owned for i in o)
        if seen != list(range(n_tiles)):
            raise RuntimeError("the ranks' tile lists do not cover the frame exactly once")
        for r in range(world):
            if r == 0 and not self_loop:
                bufs.append(local_rgba)
                continue
            nbytes = sum(tile_list[i][3] * tile_list[i][4] for i in owned[r]) * 4
            buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            bufs.append(buf)
            if nbytes:
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    if rank != 0:
        return None
    for r in range(world):
        paste_tiles(frame, bufs[r], [tile_list[i] for i in owned[r]])
    return frame


def pull_chunk(n_tiles, world, samples_per_tile, target_samples=32 << 20):
    """Tiles per pull of the dynamic queue: enough for a render call of about
    `target_samples` camera samples (small calls run the GPU half empty), but
    never so many that a rank gets fewer than two pulls."""
    want = max(1, -(-target_samples // max(1, samples_per_tile)))
    return max(1, min(want, n_tiles // max(1, 2 * world)))


def sum_counters(values, device, world, group=None, force_collective=False):
    """All-reduce of scalar counters (the frontend sums the workers' ray
    counters, Renderer.js:772-774).  Returns Python ints."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if world > 1 or force_collective:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(round(v)) for v in t.tolist()]
