"""Multi-GPU frames: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The path shards with no exchange during rendering: every pixel sample is independent and the scene,
BVH and sky are replicated on every GPU.  The ranks' partial frames are combined ONCE per readout:

  sample sharding (BASELINE C5)  every rank renders its own global sample indices of the whole frame;
                                 reduce_samples(): one reduce(SUM) of the W x H x 3 accumulator to `dst`.
  tile sharding   (BASELINE C4)  rank r renders the tile x tile tiles t with t % world == r;
                                 gather_tiles(): every rank packs its tiles into one compact slab
                                 (pt_shard_pack), ONE gather of world equal slabs to `dst`, which
                                 un-swizzles them into its frame (pt_shard_unpack).  Each rank sends only
                                 its own W*H*12/world bytes; pixel sets are disjoint, so the frame is
                                 bit-identical to the 1-GPU frame.

The reference is a single process (src/main.rs:113-170 shares one accumulator between threads); this
module has no counterpart there.  torch is plumbing here: device memory for the slabs and the collective.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import ptcore

if ptcore._lib is not None and not ptcore._torch_first:
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so); libptcore.so binds to whichever copy the process
    # loaded first.  Loaded before torch it pulls in /opt/rocm's copy, torch then brings a second one, and the second
    # runtime in a process finds no GPU ("No HIP GPUs are available").  Loaded after torch, both share torch's copy.
    raise ImportError("pathtracer_amd.dist: libptcore.so was loaded before torch was imported; import torch (or "
                      "pathtracer_amd.dist) before the first pathtracer_amd.ptcore call so both share one HIP runtime")

TILE = 64  # edge of the tiles dealt round-robin to ranks


def sample_begin(step: int, rank: int, world: int, spp: int) -> int:
    """Global index of the first sample rank `rank` renders in step `step`: step-major, then rank, so the
    union over ranks and steps is exactly the sample indices [0, steps * world * spp)."""
    return (step * world + rank) * spp


def tile_shard(rank: int, world: int, tile: int = TILE, first_sample: int = 0) -> ptcore.Shard:
    return ptcore.Shard(ptcore.SHARD_TILES, rank, world, tile, first_sample)


def slab_pixel_index(width, height, tile, rank, nranks) -> np.ndarray:
    """Linear pixel index (y * width + x) of every slot of a tile shard's slab, -1 for the padding pixels
    of edge tiles: the slab order of pt_shard_pack (tile by tile, 8x8 blocks inside a tile, row-major
    inside a block) restated in numpy, for hosts that pack on the CPU and for the tests."""
    tiles_x = (width + tile - 1) // tile
    tiles_y = (height + tile - 1) // tile
    n_tiles = tiles_x * tiles_y
    n_local = (n_tiles - rank + nranks - 1) // nranks if n_tiles > rank else 0
    idx = np.arange(n_local * tile * tile, dtype=np.int64)
    per_tile = tile * tile
    local_tile, within = idx // per_tile, idx % per_tile
    t = local_tile * nranks + rank
    tx, ty = t % tiles_x, t // tiles_x
    sub, lane = within >> 6, within & 63
    subs = tile >> 3
    x = tx * tile + (sub % subs) * 8 + (lane & 7)
    y = ty * tile + (sub // subs) * 8 + (lane >> 3)
    return np.where((x < width) & (y < height), y * width + x, -1)


def _host_staged() -> bool:
    """gloo has no device-side reduce / gather: ranks that rehearse the N > 1 path on one GPU (bench.py --backend gloo,
    tests) stage the collective through host memory.  The production backend is nccl (RCCL), device to device."""
    return dist.get_backend() == "gloo"


def gather_slabs(slab: torch.Tensor, dst: int = 0):
    """ONE gather of equally sized slabs to `dst`; returns the list of slabs there (None elsewhere)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if slab.is_cuda and _host_staged():
        host = slab.cpu()
        parts = [torch.empty_like(host) for _ in range(world)] if rank == dst else None
        dist.gather(host, parts, dst=dst)
        return [p.to(slab.device) for p in parts] if rank == dst else None
    parts = [torch.empty_like(slab) for _ in range(world)] if rank == dst else None
    dist.gather(slab, parts, dst=dst)
    return parts


def reduce_samples(pt: ptcore.PathTracer, spp_of_other_ranks: float, dst: int = 0):
    """Sample sharding: sum every rank's accumulator into `dst`'s (zero copy on the library's own device
    memory) and tell `dst` how many samples the others contributed."""
    dev = torch.device("cuda", torch.cuda.current_device())
    acc = torch.as_tensor(ptcore.AccumView(pt), device=dev)  # waits for queued render calls
    if _host_staged():
        host = acc.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            acc.copy_(host)
    else:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    if dist.get_rank() == dst:
        pt.add_external_samples(spp_of_other_ranks)


def gather_tiles(pt: ptcore.PathTracer, tile: int = TILE, dst: int = 0):
    """Tile sharding: after every rank rendered its tiles, assemble the whole frame in `dst`'s accumulator."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device())
    n = ptcore.shard_slab_floats(pt.width, pt.height, tile, 0, world)  # rank 0 owns the most tiles: common slab size
    slab = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    pt.shard_pack(tile_shard(rank, world, tile), slab.data_ptr(), n)
    parts = gather_slabs(slab, dst)
    torch.cuda.synchronize()
    if rank == dst:
        for r, part in enumerate(parts):
            if r != dst:
                pt.shard_unpack(tile_shard(r, world, tile), part.data_ptr(), n)
