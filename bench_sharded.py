"""BASELINE config 4's extra measurements for bench.py --gpus N (every rank calls the first two, in the same order; rank 0 the
third, after the process group is gone): the kernels alone (eager windows, and the same windows replayed as hipGraphs), the
"replicate" point (every GPU propagates the full catalog itself, zero bytes moved) and the one-process N-device host-returning
call (azh_group_propagate_host).  Timed like the headline: barrier, HIP events on the launch stream, max over ranks."""
import time


def _timed(torch, dist, fn, steps, stream, cuda, before=None, after=None):
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    if before:
        before()
    for _ in range(steps):
        fn()
    if after:
        after()
    e1.record(stream)
    torch.cuda.synchronize()
    t = torch.tensor([e0.elapsed_time(e1) / steps], dtype=torch.float64, device=cuda)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def kernel_only(torch, dist, a, dev, sp, plan, step, drain, stream, cuda):
    """(t_kernel_ms eager, t_kernel_graphs_ms or None): the shard's window launches without the gathers."""
    fork = lambda: sp.compute.wait_stream(stream)
    eager = _timed(torch, dist, lambda: step(False), a.steps, stream, cuda, before=fork, after=drain)
    graphs = None
    if plan.n_chunks > 1:
        # the same windows replayed as hipGraphs (azh_set_graphs: every window's launch set is one hipGraphLaunch)
        dev.set_graphs(True)
        for _ in range(4):
            step(False)       # (eager, captured, replayed x2: both redo-counter parities)
        drain()
        graphs = _timed(torch, dist, lambda: step(False), a.steps, stream, cuda, before=fork, after=drain)
        dev.set_graphs(False)
    return eager, graphs


def replicate(torch, dist, _native, synth, a, allp, sp, times, vel_on, stream, sptr, cuda, local_rank):
    """t_replicate_ms: every GPU propagates the FULL catalog itself into the gathered arrays' storage (the alternative DESIGN.md 6
    recommends to consumers that need everything everywhere)."""
    dev_full = _native.DeviceConstellation.from_tle_lines(allp, _native.WGS72, local_rank)
    dev_full.set_timing(False)
    offs_full = (synth.START_JD - dev_full.epochs) * 1440.0
    fp, fv = sp.full[0].data_ptr(), (sp.full[1].data_ptr() if vel_on else None)   # (padded >= n_total) x n_times x 3
    dev_full.propagate_device(times, offs_full, fp, fv, layout=_native.SAT_MAJOR, stream=sptr)
    run = lambda: dev_full.propagate_device_cached(fp, fv, layout=_native.SAT_MAJOR, stream=sptr)
    for _ in range(max(5, a.warmup // 4)):
        run()
    return _timed(torch, dist, run, a.steps, stream, cuda)


def sharded_screen(torch, dist, _native, synth, a, allp, plan_world, rank, times, stream, sptr, cuda, local_rank):
    """config.sharded_screen: the gather-free consumer (SURVEY 8e fallback row) -- the fused single-target screen
    (Constellation.screenConstellation, src/Constellation.zig L683-756) with the catalog sharded block-cyclically over the ranks
    (astroz_amd.distributed.ShardedScreen): every rank computes the target's track itself and screens ITS rows, no collective
    inside the step; timed like the headline (barrier, HIP events on the launch stream, max over ranks).  `value` = satellites
    x grid points screened per second over all ranks.  Checked: every rank's rows against the oracle's screen of THOSE rows and the
    target (grid indices equal, distances to 1e-6 km), gathered to rank 0 as one small object."""
    from astroz_amd.distributed import ShardedConstellation, ShardedScreen
    import numpy as np
    from oracle import oracle
    world = plan_world
    n_total = len(allp)
    target, thr = n_total // 3, 50.0
    # Every rank reaches every collective below whatever happens on it: a failure anywhere is agreed on FIRST (one all_reduce),
    # and then all ranks skip the timed part together -- an exception on one rank must not leave the others in a barrier.
    err, scr, off = None, None, None
    try:
        sh = ShardedConstellation(allp, _native.WGS72, rank=rank, world_size=world, local_rank=local_rank, n_chunks=1)
        tgt = _native.DeviceConstellation.from_tle_lines([allp[target]], _native.WGS72, local_rank)
        for d in (sh.dev, tgt):
            if d is not None:
                d.set_timing(False)
        # offsets of the whole catalog from the element sets' epochs (host-side text parsing only)
        epochs = np.array([oracle.parse_lines(a_, b_).epoch_jd for a_, b_ in allp])
        off = (synth.START_JD - epochs) * 1440.0
        scr = ShardedScreen(sh, target, tgt, times, off, thr, device=cuda)
        for _ in range(max(5, a.warmup // 4)):
            scr.step(stream=sptr)
        torch.cuda.synchronize()
    except Exception as exc:
        err = repr(exc)
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=cuda)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok[0]) == 0:
        return {"failed": err or "another rank failed during set-up"}
    ms = _timed(torch, dist, lambda: scr.step(stream=sptr), a.steps, stream, cuda)
    try:
        rows, dl, tl = scr.local_results()
        torch.cuda.synchronize()
        # the oracle's screen of THIS rank's rows against the target (sub-catalog: the target first, then the rank's other rows)
        others = np.array([r for r in rows if r != target], dtype=np.int64)
        sub = oracle.Catalog.from_pairs([allp[target]] + [allp[i] for i in others], oracle.WGS72)
        d_sub, t_sub = sub.screen_target(times, 0, thr, np.concatenate([[off[target]], off[others]]))
        d0 = np.full(n_total, thr)
        t0 = np.zeros(n_total, dtype=np.int64)
        d0[others], t0[others] = d_sub[1:], t_sub[1:]
        mine = {"rank": rank, "rows": int(len(rows)), "index_mismatches": int((tl.cpu().numpy().astype("int64") != t0[rows]).sum()),
                "max_dd_km": float(abs(dl.cpu().numpy() - d0[rows]).max()) if len(rows) else 0.0,
                "closer_than_threshold": int((dl.cpu().numpy() < thr).sum())}
    except Exception as exc:
        mine = {"rank": rank, "failed": repr(exc)}
    certs = [None] * dist.get_world_size()
    dist.all_gather_object(certs, mine)
    good = [c for c in certs if isinstance(c, dict) and "failed" not in c]
    res = {"ms": ms, "value": n_total * len(times) / (ms / 1e3), "unit": "satellite-steps screened/s", "target_row": target, "threshold_km": thr,
           "per_rank": certs,
           "what": "fused single-target screen, catalog sharded block-cyclically, target track computed on every rank, no collective in the step"}
    if len(good) == len(certs):
        res.update({"index_mismatches": sum(c["index_mismatches"] for c in good), "max_dd_km": max(c["max_dd_km"] for c in good)})
    else:
        res["check_failed_on_ranks"] = [c.get("rank") for c in certs if isinstance(c, dict) and "failed" in c]
    return res


def group_host(_native, synth, allp, world, local_rank, times, vel_on, n_total, n_times):
    """(result dict, stuck): azh_group_propagate_host on all N devices of this process, fresh host arrays each call, wall clock.
    Behind a watchdog: this is an extra, it must never take the headline line down with it."""
    import threading
    box = {}

    def work():
        try:
            text = "\n".join(x + "\n" + y for x, y in allp)
            devs = list(range(world)) if world > 1 else [local_rank]
            grp = _native.DeviceGroup(text, devs, _native.WGS72, n_chunks=1)
            goff = (synth.START_JD - grp.epochs) * 1440.0
            ws = []
            gp = gv = None
            for _ in range(5):
                del gp, gv              # (the previous result goes back to the pool outside the timed call)
                t0 = time.perf_counter()
                res_ = grp.propagate_host(times, goff, velocities=vel_on)
                ws.append((time.perf_counter() - t0) * 1e3)
                gp, gv = res_[0], res_[1]
                del res_
            gms = sorted(ws[1:])[len(ws[1:]) // 2]
            box["r"] = {"ms_per_call": gms, "calls_ms": ws, "devices": len(devs), "value": n_total * n_times / (gms / 1e3),
                        "GB_per_s": (gp.nbytes + (gv.nbytes if gv is not None else 0)) / (gms / 1e3) / 1e9,
                        "what": "azh_group_propagate_host: one process, N devices, fresh (pinned, pooled) host arrays each call; wall clock"}
            del gp, gv
            grp.close()
        except Exception as exc:
            box["r"] = {"failed": repr(exc)}
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout=120.0)
    return box.get("r", {"failed": "timed out after 120 s"}), th.is_alive()
