"""Multi-GPU: one process per GPU, satellites sharded across ranks.

The path shards embarrassingly -- every (satellite, time) result is independent for SGP4, and the
only dependency of SDP4 runs along time inside one satellite -- so there is NO data-path
collective: each rank owns a contiguous range of the catalog, builds its own device-resident
element table and writes its own block of the result.  The reference has no distributed layer
at all (single process, std.Thread: src/Constellation.zig L327-385).

The one optional exchange is re-assembling the full ``(n_sats, n_times, 3)`` array on every rank
(``gather_sat_major``): a single RCCL all-gather over xGMI of blocks that are contiguous in a
satellite-major layout.  Payload at 13,478 x 1,440 fp64 pos+vel is 931.6 MB per rank in total
(116.4 MB contributed per rank at 8 GPUs), ~5 ms on a ring -- 10x the kernel -- so consumers that
can work on their own shard (conjunction screening, ground-track products) should skip it.
"""
import os

import numpy as np

SHARD_ALIGN = 64  # one wave of satellites


def shard_bounds(n_sats, world_size, rank, align=SHARD_ALIGN):
    """Contiguous [lo, hi) of the catalog owned by `rank`: ceil(n/world) rounded up to `align`
    satellites per rank, last ranks possibly short or empty."""
    per = -(-n_sats // world_size)
    per = -(-per // align) * align
    lo = min(rank * per, n_sats)
    hi = min(lo + per, n_sats)
    return lo, hi


def shard_sizes(n_sats, world_size, align=SHARD_ALIGN):
    return [shard_bounds(n_sats, world_size, r, align)[1] - shard_bounds(n_sats, world_size, r, align)[0]
            for r in range(world_size)]


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


class ShardedConstellation:
    """The rank's slice of a TLE catalog on the rank's GPU.

    `pairs` is the FULL catalog (every rank passes the same list); only the rank's contiguous
    range is parsed, uploaded and initialised."""

    def __init__(self, pairs, grav=1, rank=None, world_size=None, local_rank=None):
        from . import _native

        r, w, lr = env_rank()
        self.rank = r if rank is None else rank
        self.world_size = w if world_size is None else world_size
        self.local_rank = lr if local_rank is None else local_rank
        self.n_total = len(pairs)
        self.lo, self.hi = shard_bounds(self.n_total, self.world_size, self.rank)
        self.dev = None
        if self.hi > self.lo:
            self.dev = _native.DeviceConstellation.from_tle_lines(pairs[self.lo:self.hi], grav, self.local_rank)

    @property
    def n_local(self):
        return self.hi - self.lo


def gather_sat_major(local_block, n_total, world_size=None, group=None):
    """All-gather satellite-major blocks ``(n_local, n_times, 3)`` into ``(n_total, n_times, 3)``.

    Works for CPU tensors (gloo) and GPU tensors (nccl = RCCL).  Ranks may own different numbers
    of satellites (the tail ranks are short), so blocks are padded to the common shard size for
    the collective and trimmed afterwards; with equal shards this is a single
    all_gather_into_tensor straight into the output."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if world_size is None else world_size
    sizes = shard_sizes(n_total, world)
    per = max(sizes)
    n_times = local_block.shape[1]
    if all(s == per for s in sizes):
        out = torch.empty((world * per, n_times, 3), dtype=local_block.dtype, device=local_block.device)
        dist.all_gather_into_tensor(out, local_block.contiguous(), group=group)
        return out
    padded = torch.zeros((per, n_times, 3), dtype=local_block.dtype, device=local_block.device)
    padded[: local_block.shape[0]] = local_block
    flat = torch.empty((world * per, n_times, 3), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(flat, padded, group=group)
    buf = flat.view(world, per, n_times, 3)
    return torch.cat([buf[r, : sizes[r]] for r in range(world)], dim=0)


def gather_time_major(local_block, n_total, world_size=None, group=None):
    """Same for time-major blocks ``(n_times, n_local, 3)`` -> ``(n_times, n_total, 3)``: each
    rank owns a column block, so the gathered buffer is permuted once after the collective."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if world_size is None else world_size
    sizes = shard_sizes(n_total, world)
    per = max(sizes)
    n_times = local_block.shape[0]
    padded = local_block
    if local_block.shape[1] != per:
        padded = torch.zeros((n_times, per, 3), dtype=local_block.dtype, device=local_block.device)
        padded[:, : local_block.shape[1]] = local_block
    flat = torch.empty((world * n_times, per, 3), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(flat, padded.contiguous(), group=group)
    buf = flat.view(world, n_times, per, 3)
    return torch.cat([buf[r, :, : sizes[r]] for r in range(world)], dim=1)
