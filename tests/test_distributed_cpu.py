"""N > 1 path on CPU: world_size-2 gloo.  The sharding arithmetic and the optional all-gather
re-assembly are exercised with the ORACLE standing in for the per-rank GPU kernel (this file is a
test: it may call the oracle; the product's distributed module never does)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_catalog():
    from astroz_amd.distributed import shard_bounds, shard_sizes
    for n in (1, 63, 64, 65, 1000, 13478, 15000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            assert all((b - a) % 64 == 0 for a, b in spans[:-1] if b < n)
            assert sum(shard_sizes(n, w)) == n
    assert shard_bounds(13478, 8, 0) == (0, 1728) and shard_bounds(13478, 8, 7) == (12096, 13478)


def test_shard_plan_block_cyclic():
    from astroz_amd.distributed import ShardPlan
    for n in (1, 63, 64, 65, 1000, 13478, 15000):
        for w in (1, 2, 3, 4, 8):
            for c in (1, 2, 4, 7):
                pl = ShardPlan(n, w, c)
                rows = [pl.local_rows(r) for r in range(w)]
                allr = np.sort(np.concatenate(rows))
                assert np.array_equal(allr, np.arange(n)), (n, w, c)            # a partition of the catalog
                assert pl.rows % 64 == 0 and pl.padded == pl.n_chunks * w * pl.rows >= n
                for r in range(w):
                    assert np.all(np.diff(rows[r]) > 0) and len(rows[r]) == pl.n_local(r)
                    # local index = chunk*rows + slot: windows tile the local block, in order
                    at = 0
                    for k in range(pl.n_chunks):
                        lo, hi = pl.chunk_window(k, r)
                        assert lo == k * pl.rows and hi - lo == len(np.arange(*pl.cell(k, r)))
                        if hi > lo:
                            assert lo == at or at % pl.rows == 0
                            assert np.array_equal(rows[r][at:at + hi - lo], np.arange(*pl.cell(k, r)))
                            assert lo == at     # full cells first, then at most one partial cell
                        at += hi - lo
    # a catalog sorted by regime (deep-space members last) still loads every rank evenly
    # (as long as a super-block, world*rows satellites, is short against the expensive run: use more chunks)
    pl = ShardPlan(15000, 8, 16)
    deep = np.arange(15000) >= 13478
    per_rank = [int(deep[pl.local_rows(r)].sum()) for r in range(8)]
    assert max(per_rank) - min(per_rank) <= pl.rows and min(per_rank) > 0, per_rank
    contiguous = [int(deep[r * 1875:(r + 1) * 1875].sum()) for r in range(8)]      # plain ranges: all on the last rank
    assert max(contiguous) == 1522 and max(per_rank) <= 2 * pl.rows


class _OracleDevice:
    """Stand-in for DeviceConstellation in the CPU tier: same propagate_device_window contract, the oracle does
    the arithmetic (this file is a test; the product's distributed module never touches the oracle)."""

    def __init__(self, cat, times, off, capacity):
        self.cat, self.times, self.off, self.cap = cat, times, off, capacity
        self.calls = []

    def propagate_device_window(self, lo, hi, d_pos, d_vel, layout=0, stream=None):
        import ctypes as C
        from oracle import oracle
        assert layout == 0 and stream is None
        nt = len(self.times)
        _, p, v = self.cat.propagate(self.times, self.off, layout=oracle.SAT_MAJOR)
        hi = min(hi, self.cat.n)
        self.calls.append((lo, hi))
        for ptr, src in ((d_pos, p), (d_vel, v)):
            if ptr:
                dst = np.ctypeslib.as_array((C.c_double * (self.cap * nt * 3)).from_address(ptr)).reshape(self.cap, nt, 3)
                dst[lo:hi] = src[lo:hi]


def _worker(rank, world, port, n_near, n_deep, n_chunks, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from astroz_amd import synth
    from astroz_amd.distributed import ShardPlan, ShardedPropagator, gather_sat_major, gather_time_major, shard_bounds
    from oracle import oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = synth.synth_catalog(n_near, n_deep, seed=31, interleave=False)   # sorted by regime on purpose
        n = len(pairs)
        times = np.arange(0.0, 90.0, 3.0)
        ref = oracle.Catalog.from_pairs(pairs, 1)
        roff = (synth.START_JD - ref.epoch_jd) * 1440.0
        _, r_sm, rv_sm = ref.propagate(times, roff, layout=oracle.SAT_MAJOR)
        # (1) the config-4 pipeline of bench.py: block-cyclic plan, chunk windows, chunked all-gather
        plan = ShardPlan(n, world, n_chunks, align=8)
        mine = plan.local_rows(rank)
        cat = oracle.Catalog.from_pairs([pairs[i] for i in mine], 1)      # stand-in for the rank's GPU shard
        off = (synth.START_JD - cat.epoch_jd) * 1440.0
        dev = _OracleDevice(cat, times, off, plan.local_capacity())
        sp = ShardedPropagator(dev, plan, rank, len(times), velocities=True)
        sp.step(gather=True)
        sp.wait()
        pos, vel = [t.numpy() for t in sp.results()]
        ok = pos.shape == r_sm.shape and np.array_equal(pos, r_sm) and np.array_equal(vel, rv_sm)
        ok = ok and len(dev.calls) == sum(1 for c in range(plan.n_chunks) if plan.chunk_window(c, rank)[1] > plan.chunk_window(c, rank)[0])
        rows, loc = sp.local_results()
        ok = ok and np.array_equal(rows, mine) and np.array_equal(loc[0].numpy(), r_sm[mine])
        # (2) the one-shot gathers of contiguous-range shards
        lo, hi = shard_bounds(n, world, rank)
        ccat = oracle.Catalog.from_pairs(pairs[lo:hi], 1)
        coff = (synth.START_JD - ccat.epoch_jd) * 1440.0
        _, p_sm, _ = ccat.propagate(times, coff, layout=oracle.SAT_MAJOR, velocities=False)
        _, p_tm, _ = ccat.propagate(times, coff, layout=oracle.TIME_MAJOR, velocities=False)
        full_sm = gather_sat_major(torch.from_numpy(p_sm), n).numpy()
        full_tm = gather_time_major(torch.from_numpy(p_tm), n).numpy()
        ok = ok and (full_sm.shape == r_sm.shape and np.array_equal(full_sm, r_sm)
                     and np.array_equal(full_tm, r_sm.transpose(1, 0, 2)))
        q.put((rank, bool(ok), int(len(mine)), int(cat.is_deep.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_near,n_deep,n_chunks", [(256, 0, 1), (150, 37, 3), (333, 40, 4)])
def test_gloo_world2_config4_pipeline(n_near, n_deep, n_chunks):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_near, n_deep, n_chunks, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2] + res[1][2] == n_near + n_deep
    if n_deep and n_chunks > 1:
        assert res[0][3] > 0 and res[1][3] > 0, res      # deep-space members (sorted last) reach both ranks


# ---- the gather-free consumer: the sharded fused screen (SURVEY 8e fallback row) --------------------------------------------
def _as_np(ptr, shape, ctype, dtype):
    import ctypes as C
    n = int(np.prod(shape))
    return np.ctypeslib.as_array((ctype * n).from_address(ptr)).view(dtype).reshape(shape)


class _OracleTarget:
    """stand-in for the one-satellite DeviceConstellation of the target (propagate_one_device contract, host pointers)"""

    def __init__(self, cat):
        self.cat = cat

    def propagate_one_device(self, sat, d_tsince, n, d_pos, d_vel=None, d_err=None, stream=None):
        import ctypes as C
        ts = _as_np(d_tsince, (n,), C.c_double, np.float64)
        pos = _as_np(d_pos, (n, 3), C.c_double, np.float64)
        err = _as_np(d_err, (n,), C.c_uint8, np.uint8) if d_err else None
        for k in range(n):
            rc, r, _ = self.cat.propagate_one(sat, ts[k])
            pos[k] = r if rc == 0 else 0.0
            if err is not None:
                err[k] = rc

    def synchronize(self):
        pass


class _OracleScreenShard:
    """stand-in for the rank's DeviceConstellation: screen_track_device with the oracle's positions"""

    def __init__(self, cat):
        self.cat, self.n = cat, cat.n

    def screen_track_device(self, times, d_track, threshold, d_min_dist, d_min_t, offsets_min=None, exclude=None, stream=None):
        import ctypes as C
        from oracle import oracle
        nt = len(times)
        track = _as_np(d_track, (nt, 3), C.c_double, np.float64)
        out_d = _as_np(d_min_dist, (self.n,), C.c_double, np.float64)
        out_t = _as_np(d_min_t, (self.n,), C.c_int32, np.int32)
        err, p, _ = self.cat.propagate(times, offsets_min, layout=oracle.SAT_MAJOR, velocities=False)
        d2 = ((p - track[None]) ** 2).sum(axis=2)
        d2[err != 0] = np.inf
        d2[:, ~np.isfinite(track).all(axis=1)] = np.inf
        for s in range(self.n):
            best, bt = threshold * threshold, 0
            if s != exclude and self.cat.init_rc[s] == 0:
                k = int(np.argmin(d2[s]))
                if d2[s, k] < best:
                    best, bt = d2[s, k], k
            out_d[s] = np.sqrt(best)
            out_t[s] = bt


def _screen_worker(rank, world, port, n_near, n_deep, n_chunks, target, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from astroz_amd import synth
    from astroz_amd.distributed import ShardPlan, ShardedScreen
    from oracle import oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = synth.synth_catalog(n_near, n_deep, seed=17)
        n = len(pairs)
        times = np.arange(0.0, 240.0, 2.0)
        ref = oracle.Catalog.from_pairs(pairs, 1)
        roff = (synth.START_JD - ref.epoch_jd) * 1440.0
        thr = 2500.0
        d0, t0 = ref.screen_target(times, target, thr, roff)

        class Shard:
            pass
        sh = Shard()
        sh.plan = ShardPlan(n, world, n_chunks, align=8)
        sh.rank = rank
        sh.rows = sh.plan.local_rows(rank)
        sh.dev = _OracleScreenShard(oracle.Catalog.from_pairs([pairs[i] for i in sh.rows], 1)) if len(sh.rows) else None
        scr = ShardedScreen(sh, target, _OracleTarget(oracle.Catalog.from_pairs([pairs[target]], 1)), times, roff, thr)
        scr.step()
        rows, dl, tl = scr.local_results()
        ok = np.array_equal(rows, sh.rows) and np.allclose(dl.numpy(), d0[rows], rtol=0, atol=1e-9) and np.array_equal(tl.numpy(), t0[rows])
        ok = ok and ((scr.exclude is not None) == (target in set(int(x) for x in rows)))
        dg, tg = scr.gather()
        ok = ok and np.allclose(dg.numpy(), d0, rtol=0, atol=1e-9) and np.array_equal(tg.numpy(), t0)
        q.put((rank, bool(ok), int((d0 < thr).sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_near,n_deep,n_chunks,target", [(200, 0, 1, 3), (150, 30, 3, 170), (97, 11, 2, 64)])
def test_gloo_world2_sharded_screen(n_near, n_deep, n_chunks, target):
    """every rank screens its block-cyclic rows against the target's track (computed by every rank itself): local results =
    the oracle's screen restricted to the rank's rows, the 12-byte-per-satellite gather = the oracle's whole result; no
    collective inside the step"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_screen_worker, args=(r, 2, port, n_near, n_deep, n_chunks, target, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2] > 1, res      # (the threshold is wide enough that the screen has something to find)
