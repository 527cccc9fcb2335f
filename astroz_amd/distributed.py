"""Multi-GPU: one process per GPU, satellites sharded across ranks (BASELINE config 4).

The path shards embarrassingly -- every (satellite, time) result is independent for SGP4, and the only
dependency of SDP4 runs along time inside one satellite -- so the propagation itself needs no collective:
each rank owns part of the catalog, builds its own device-resident element table and writes its own
block of the result.  The reference has no distributed layer at all (single process, std.Thread:
src/Constellation.zig L327-385).

The one exchange is re-assembling the full ``(n_sats, n_times, 3)`` array on every rank: RCCL
all-gathers over xGMI (``torch.distributed`` backend "nccl" on ROCm).  Two things shape it:

* **Block-cyclic ownership** (:class:`ShardPlan`).  The catalog is cut into ``n_chunks`` super-blocks of
  ``world * rows`` consecutive satellites; inside every super-block rank r owns rows
  ``[r*rows, (r+1)*rows)``.  A rank's shard is then ``n_chunks`` runs of consecutive satellites, and the
  all-gather of chunk c lands in ``out[c*world*rows : (c+1)*world*rows]`` -- contiguous, already in
  catalog order, no repacking pass.  Because every super-block is spread over all ranks, a catalog
  that is sorted by orbit regime (all deep-space members at the end: 2-3x the cost per propagation)
  still loads the ranks evenly once a super-block is short against that run (1,522 deep-space members at
  the end of 15,000: 16 chunks give every one of 8 ranks 128-256 of them; plain contiguous ranges put all
  1,522 on the last rank).
* **Chunk pipeline** (:class:`ShardedPropagator`).  Chunk c+1 is computed
  (``azh_propagate_device_window``) while chunk c is in flight on the communication stream.

Cost model: the gather moves ``(world-1)/world`` of the 931.6 MB result into every GPU; an
MI355X ingests at most 7 x ~75 GB/s over xGMI, so t_allgather >= 1.6 ms at 8 GPUs (>= 6 ms at 2, one
link) against 0.26 ms to compute the WHOLE result on one GPU.  Consumers that can work on their own shard
(conjunction screening, ground tracks) should skip the gather; consumers that need everything everywhere
are better served by every GPU propagating the full catalog itself -- 0.2 ms, and no bytes move at all.
"""
import os

import numpy as np

SHARD_ALIGN = 64  # one wave of satellites


# ----------------------------------------------------------------------------------------------
# contiguous ranges (kept for callers that want plain [lo, hi) shards and no gather pipeline)
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


# ----------------------------------------------------------------------------------------------
class ShardPlan:
    """Block-cyclic assignment of `n_sats` catalog rows to `world` ranks in `n_chunks` super-blocks.

    rows   : satellites per (chunk, rank) cell, a multiple of `align`
    padded : n_chunks * world * rows >= n_sats -- the row count of the gathered buffer; rows
             ``[n_sats, padded)`` are padding
    catalog row g  ->  chunk g // (world*rows), rank (g % (world*rows)) // rows, slot g % rows
    A rank's members, in ascending catalog order, fill its local block chunk after chunk: full cells
    first, then at most one partial cell, then empty ones -- so local index l = chunk*rows + slot always.
    """

    def __init__(self, n_sats, world, n_chunks=1, align=SHARD_ALIGN):
        if n_sats < 0 or world < 1 or n_chunks < 1:
            raise ValueError("bad shard plan arguments")
        self.n_sats, self.world = int(n_sats), int(world)
        cells = self.world * int(n_chunks)
        rows = -(-max(self.n_sats, 1) // cells)
        self.rows = -(-rows // align) * align
        # no point in chunks that are entirely padding
        self.n_chunks = max(1, min(int(n_chunks), -(-max(self.n_sats, 1) // (self.world * self.rows))))
        self.padded = self.n_chunks * self.world * self.rows

    def cell(self, chunk, rank):
        """[lo, hi) catalog rows of (chunk, rank), clipped to the catalog."""
        lo = (chunk * self.world + rank) * self.rows
        return min(lo, self.n_sats), min(lo + self.rows, self.n_sats)

    def local_rows(self, rank):
        """catalog indices owned by `rank`, ascending (== local order)."""
        parts = [np.arange(*self.cell(c, rank), dtype=np.int64) for c in range(self.n_chunks)]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)

    def n_local(self, rank):
        return sum(hi - lo for lo, hi in (self.cell(c, rank) for c in range(self.n_chunks)))

    def chunk_window(self, chunk, rank):
        """[lo, hi) LOCAL rows of `rank` that belong to `chunk`."""
        lo, hi = self.cell(chunk, rank)
        return chunk * self.rows, chunk * self.rows + (hi - lo)

    def local_capacity(self):
        """rows of a rank's local block buffer (cells are padded to `rows`)."""
        return self.n_chunks * self.rows


class ShardedConstellation:
    """The rank's block-cyclic shard of a TLE catalog on the rank's GPU.

    `pairs` is the FULL catalog (every rank passes the same list); only the rank's members are
    parsed, uploaded and initialised."""

    def __init__(self, pairs, grav=1, rank=None, world_size=None, local_rank=None, n_chunks=1):
        from . import _native

        r, w, lr = env_rank()
        self.rank = r if rank is None else rank
        self.world_size = w if world_size is None else world_size
        self.local_rank = lr if local_rank is None else local_rank
        self.n_total = len(pairs)
        self.plan = ShardPlan(self.n_total, self.world_size, n_chunks)
        self.rows = self.plan.local_rows(self.rank)
        self.dev = None
        if len(self.rows):
            self.dev = _native.DeviceConstellation.from_tle_lines([pairs[i] for i in self.rows], grav, self.local_rank)

    @property
    def n_local(self):
        return len(self.rows)


class ShardedPropagator:
    """Config 4: every rank propagates its shard and all ranks end up with the full satellite-major
    ``(n_sats, n_times, 3)`` arrays, chunk-pipelined (compute of chunk c+1 overlaps the gather of chunk c).

    `dev` is anything with ``propagate_device_window(row_lo, row_hi, d_pos, d_vel, layout=, stream=)``
    writing rows [row_lo, row_hi) of satellite-major local blocks -- a
    :class:`astroz_amd._native.DeviceConstellation` whose inputs have been staged by one
    ``propagate_device`` call, or a stand-in in the CPU tests.  Works on CUDA tensors (backend nccl =
    RCCL) and on CPU tensors (gloo; no streams, chunks run back to back)."""

    def __init__(self, dev, plan, rank, n_times, velocities=True, device=None, dtype=None, group=None):
        import torch

        self.torch = torch
        self.dev, self.plan, self.rank, self.n_times = dev, plan, int(rank), int(n_times)
        self.group = group
        self.device = torch.device("cpu") if device is None else device
        self.cuda = self.device.type == "cuda"
        dtype = torch.float64 if dtype is None else dtype
        n_arr = 2 if velocities else 1
        cap = plan.local_capacity()
        # local blocks (what the kernels write) and the gathered, catalog-ordered result
        self.local = [torch.zeros((cap, n_times, 3), dtype=dtype, device=self.device) for _ in range(n_arr)]
        self.full = [torch.empty((plan.padded, n_times, 3), dtype=dtype, device=self.device) for _ in range(n_arr)]
        self._ordered = False
        if self.cuda:
            self.compute = torch.cuda.Stream(device=self.device)
            self.comm = torch.cuda.Stream(device=self.device)
            self.ev_chunk = [torch.cuda.Event() for _ in range(plan.n_chunks)]

    # -- pieces ---------------------------------------------------------------------------------
    def _launch_chunk(self, c):
        lo, hi = self.plan.chunk_window(c, self.rank)
        if hi <= lo or self.dev is None:
            return
        p = self.local[0].data_ptr()
        v = self.local[1].data_ptr() if len(self.local) > 1 else None
        stream = self.compute.cuda_stream if self.cuda else None
        self.dev.propagate_device_window(lo, hi, p, v, layout=0, stream=stream)

    def _gather_chunk(self, c):
        import torch.distributed as dist

        rows, w = self.plan.rows, self.plan.world
        for loc, full in zip(self.local, self.full):
            dist.all_gather_into_tensor(full[c * w * rows:(c + 1) * w * rows], loc[c * rows:(c + 1) * rows],
                                        group=self.group)

    # -- one step --------------------------------------------------------------------------------
    def step(self, gather=True):
        """Propagate the shard (all chunks) and, if `gather`, all-gather every chunk.  Asynchronous on
        CUDA: call :meth:`wait` (or synchronize) before reading ``results()``."""
        torch = self.torch
        if self.cuda and not self._ordered:
            # first step: three streams have touched what the kernels are about to use -- the zero-fill of the local
            # blocks (torch's current stream at construction), the input staging of the caller's propagate_device call
            # (the handle's own stream or the one the caller passed) and this object's compute stream.  One device-wide
            # wait orders them all; later steps are ordered by the stream waits below.
            torch.cuda.synchronize(self.device)
            self._ordered = True
        if self.cuda:
            # the next step's kernels must not overwrite local blocks the previous gathers still read
            self.compute.wait_stream(self.comm)
        for c in range(self.plan.n_chunks):
            self._launch_chunk(c)
            if not gather:
                continue
            if self.cuda:
                self.ev_chunk[c].record(self.compute)
                self.comm.wait_event(self.ev_chunk[c])
                with torch.cuda.stream(self.comm):
                    self._gather_chunk(c)
            else:
                self._gather_chunk(c)

    def wait(self):
        if self.cuda:
            self.torch.cuda.current_stream(self.device).wait_stream(self.comm)
            self.torch.cuda.current_stream(self.device).wait_stream(self.compute)

    def results(self):
        """Catalog-ordered views ``(n_sats, n_times, 3)`` of the gathered arrays (pos[, vel])."""
        return [f[: self.plan.n_sats] for f in self.full]

    def local_results(self):
        """(catalog rows of this rank, local blocks trimmed to them)."""
        rows = self.plan.local_rows(self.rank)
        return rows, [l[: len(rows)] for l in self.local]


# ----------------------------------------------------------------------------------------------
class ShardedScreen:
    """The gather-free consumer (SURVEY 8e fallback row): the fused single-target conjunction screen
    (Constellation.screenConstellation, src/Constellation.zig L683-756; ``screen(..., target=)``, bindings/python/astroz/
    __init__.py L535-658) with the catalog sharded over the ranks.  A satellite's minimum distance to the target depends on
    that satellite and the target only, so every rank screens ITS rows; the target's track -- ``n_times x 24`` bytes -- is
    computed by every rank for itself from the target's element set (``target_dev``: a one-satellite constellation on the
    rank's GPU), so the step has NO collective and its time falls as 1/world.  ``gather()`` collects the 12 bytes per
    satellite of the result on every rank when somebody needs the whole vector (162 KB for 13,478 satellites).

    `shard`: the rank's :class:`ShardedConstellation` (or anything with ``dev``, ``rows``, ``plan``, ``rank``);
    `target_row`: catalog row of the target; `target_dev`: a DeviceConstellation holding that one satellite (None on CPU
    stand-ins, whose ``screen_track_device`` takes the track as a tensor)."""

    def __init__(self, shard, target_row, target_dev, times_min, offsets_min, threshold_km, device=None, group=None):
        import torch

        self.torch, self.shard, self.group = torch, shard, group
        self.device = torch.device("cpu") if device is None else device
        self.cuda = self.device.type == "cuda"
        self.times = np.ascontiguousarray(times_min, dtype=np.float64)
        off = None if offsets_min is None else np.ascontiguousarray(offsets_min, dtype=np.float64)
        self.offsets_local = None if off is None else np.ascontiguousarray(off[shard.rows])
        self.target_off = 0.0 if off is None else float(off[target_row])
        self.threshold = float(threshold_km)
        self.target_dev = target_dev
        hit = np.flatnonzero(np.asarray(shard.rows) == int(target_row))
        self.exclude = int(hit[0]) if len(hit) else None           # the target is one of this rank's rows
        n_loc, n_t = len(shard.rows), len(self.times)
        self.track = torch.zeros((n_t, 3), dtype=torch.float64, device=self.device)
        self.tsince = torch.as_tensor(self.times + self.target_off, dtype=torch.float64).to(self.device)
        self.min_dist = torch.full((max(n_loc, 1),), self.threshold, dtype=torch.float64, device=self.device)[:n_loc]
        self.min_t = torch.zeros((max(n_loc, 1),), dtype=torch.int32, device=self.device)[:n_loc]
        self.track_err = torch.zeros((n_t,), dtype=torch.uint8, device=self.device)
        self._check_track, self._has_bad, self._bad = True, False, None
        if self.cuda:
            torch.cuda.synchronize(self.device)

    def step(self, stream=None):
        """target track + screen of the rank's rows; asynchronous on CUDA: pass ONE raw hipStream_t as `stream` so that the
        two handles' launches are ordered on it (without it the track is finished with a host synchronize first)."""
        n_t = len(self.times)
        if self.target_dev is not None:
            self.target_dev.propagate_one_device(0, self.tsince.data_ptr(), n_t, self.track.data_ptr(), None, self.track_err.data_ptr(),
                                                 stream=stream)
            if self._check_track:
                # once per grid: a target whose propagation fails at some grid points must never compare closer than the
                # threshold there (azh_screen_target_* marks such points NaN); the usual grid has none and costs nothing later
                self.target_dev.synchronize()
                if self.cuda and stream is not None:
                    self.torch.cuda.synchronize(self.device)
                self._bad = (self.track_err != 0)
                self._has_bad = bool(self._bad.any())
                self._check_track = False
            if self._has_bad:
                self.target_dev.synchronize()
                if self.cuda and stream is not None:
                    self.torch.cuda.synchronize(self.device)
                self.track[self._bad] = float("nan")
                if self.cuda:
                    self.torch.cuda.synchronize(self.device)
            elif self.cuda and stream is None:
                self.target_dev.synchronize()      # two handles, two streams: the screen must see the finished track
        if self.shard.dev is None or not len(self.shard.rows):
            return
        self.shard.dev.screen_track_device(self.times, self.track.data_ptr(), self.threshold, self.min_dist.data_ptr(),
                                           self.min_t.data_ptr(), offsets_min=self.offsets_local, exclude=self.exclude, stream=stream)

    def local_results(self):
        """(catalog rows of this rank, min_dist km, min_t_index) -- device tensors of the rank's rows"""
        return self.shard.rows, self.min_dist, self.min_t

    def gather(self):
        """catalog-ordered (min_dist (n_sats,), min_t (n_sats,)) on every rank: one all-gather of 12 bytes per satellite"""
        import torch.distributed as dist

        torch, plan = self.torch, self.shard.plan
        cap = plan.local_capacity()
        loc = torch.zeros((cap, 2), dtype=torch.float64, device=self.device)
        n_loc = len(self.shard.rows)
        # local row l of chunk c, slot s sits at c * rows + s: the cells of a rank are filled in order, so the first n_loc
        # local rows map onto (chunk, slot) by plain division
        idx = torch.arange(n_loc, device=self.device)
        loc[idx, 0] = self.min_dist
        loc[idx, 1] = self.min_t.to(torch.float64)
        full = torch.empty((plan.world * cap, 2), dtype=torch.float64, device=self.device)
        dist.all_gather_into_tensor(full, loc, group=self.group)
        full = full.view(plan.world, plan.n_chunks, plan.rows, 2).permute(1, 0, 2, 3).reshape(plan.padded, 2)[: plan.n_sats]
        return full[:, 0].contiguous(), full[:, 1].to(torch.int64)


# ----------------------------------------------------------------------------------------------
# one-shot gathers of contiguous-range shards (shard_bounds), kept for API users
def gather_sat_major(local_block, n_total, world_size=None, group=None):
    """All-gather satellite-major blocks ``(n_local, n_times, 3)`` of `shard_bounds` shards into
    ``(n_total, n_times, 3)``.  Works for CPU tensors (gloo) and GPU tensors (nccl = RCCL).  The
    blocks are padded to the common shard size for the collective; only the last ranks are short, so
    the gathered buffer trimmed to `n_total` rows is already the catalog-ordered result."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if world_size is None else world_size
    sizes = shard_sizes(n_total, world)
    per = max(sizes)
    n_times = local_block.shape[1]
    block = local_block.contiguous()
    if block.shape[0] != per:
        block = torch.zeros((per, n_times, 3), dtype=local_block.dtype, device=local_block.device)
        block[: local_block.shape[0]] = local_block
    flat = torch.empty((world * per, n_times, 3), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(flat, block, group=group)
    return flat[:n_total]


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
