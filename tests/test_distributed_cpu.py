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


def _worker(rank, world, port, n_near, n_deep, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from astroz_amd import synth
    from astroz_amd.distributed import gather_sat_major, gather_time_major, shard_bounds
    from oracle import oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = synth.synth_catalog(n_near, n_deep, seed=31)
        n = len(pairs)
        lo, hi = shard_bounds(n, world, rank)
        times = np.arange(0.0, 90.0, 3.0)
        cat = oracle.Catalog.from_pairs(pairs[lo:hi], 1)          # stand-in for the rank's GPU kernel
        off = (synth.START_JD - cat.epoch_jd) * 1440.0
        _, p_sm, _ = cat.propagate(times, off, layout=oracle.SAT_MAJOR, velocities=False)
        _, p_tm, _ = cat.propagate(times, off, layout=oracle.TIME_MAJOR, velocities=False)
        full_sm = gather_sat_major(torch.from_numpy(p_sm), n).numpy()
        full_tm = gather_time_major(torch.from_numpy(p_tm), n).numpy()
        ref = oracle.Catalog.from_pairs(pairs, 1)
        roff = (synth.START_JD - ref.epoch_jd) * 1440.0
        _, r_sm, _ = ref.propagate(times, roff, layout=oracle.SAT_MAJOR, velocities=False)
        ok = (full_sm.shape == r_sm.shape and np.array_equal(full_sm, r_sm)
              and np.array_equal(full_tm, r_sm.transpose(1, 0, 2)))
        q.put((rank, bool(ok), lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_near,n_deep", [(256, 0), (150, 37)])
def test_gloo_world2_gather(n_near, n_deep):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_near, n_deep, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == n_near + n_deep
