#!/usr/bin/env python3
"""Headline benchmark: propagations/sec, 13,478 satellites x 1,440 one-minute steps (BASELINE.json).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole workload: every satellite of the catalog propagated to
every time, fp64, TEME, positions + velocities, written into device-resident (n_sats, n_times, 3) x 2 outputs
(the shape BASELINE.json's north_star names; `--layout time` selects the (n_times, n_sats, 3) layout and
with it the lane = satellite kernel).  Inputs (element table, time grid, epoch offsets) are resident in HBM
before the timed region starts; nothing is copied to the host inside it.

N = 1 is BASELINE config 2.  N > 1 is config 4 by default: STRONG scaling of the same 13,478-satellite
catalog -- block-cyclic satellite shards (astroz_amd.distributed.ShardPlan), every rank propagates its shard and
RCCL all-gathers re-assemble the full (n_sats, n_times, 3) arrays on every GPU, chunk-pipelined so that chunk
c+1 is computed while chunk c is in flight; `value` = 13,478 x 1,440 / t_total, and `config` carries t_kernel,
t_allgather and t_total separately.  `--scaling weak` (every rank its own 13,478-satellite catalog, no gather)
and `--no-gather` are explicit alternatives and say so in `config.workload`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic traffic / work per propagation (restated in DESIGN.md)
from bench_common import (BYTES_OUT_P, BYTES_OUT_PV, ELEM_BYTES_PER_SAT, FLOPS_PER_PROP, FP64_VALU_PEAK_TF, HBM_PEAK_GBS,  # noqa: E402
                          PowerSampler, _sample_rows, compact_line, cpu_baseline, csrc_fingerprint, describe_workload, usable_cpus)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--precondition-ms", type=float, default=200.0,
                    help="run the step back to back for this long before the W warm-up steps (device clocks "
                         "reach their sustained level ~30 ms after load onset; 0 disables)")
    ap.add_argument("--sats", type=int, default=13478)
    ap.add_argument("--times", type=int, default=1440)
    ap.add_argument("--deep", type=int, default=0, help="extra deep-space satellites (config 3: 1522)")
    ap.add_argument("--pos-only", action="store_true")
    ap.add_argument("--f32-out", action="store_true",
                    help="fp32 OUTPUT arrays (BASELINE config 5); the arithmetic stays fp64")
    ap.add_argument("--layout", choices=["time", "sat"], default="sat",
                    help="physical output layout: sat = (n_sats, n_times, 3) [default], time = (n_times, n_sats, 3)")
    ap.add_argument("--mode", choices=["teme", "ecef", "geodetic"], default="teme",
                    help="output frame (SURVEY 8 f1): teme [default, BASELINE.json], ecef (the default of the reference's high-level "
                         "propagate(), Constellation.zig L489-506), geodetic (lat rad, lon rad, alt km; velocities stay ECEF)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong = one 13,478-satellite catalog sharded over the ranks (BASELINE config 4, default); "
                         "weak = every rank its own catalog, no gather")
    ap.add_argument("--no-gather", action="store_true", help="N > 1, strong: skip the RCCL all-gather of the result")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N > 1 code path (shard plan, chunk pipeline, RCCL all-gather) with whatever world size there "
                         "is, including 1: a smoke test of the config-4 path on a single-GPU box")
    ap.add_argument("--chunks", type=int, default=0, help="N > 1: pipeline depth of the compute / all-gather overlap (0 = from the rows per rank: one chunk per ~3,000 rows, at most 4)")
    ap.add_argument("--tile", type=int, default=0, help="time steps per workgroup (0 = auto)")
    ap.add_argument("--stride-align", type=int, default=0,
                    help="time-major only: round the row length (out_stride_sats) up to a multiple of this many "
                         "satellites (16 = 128-byte aligned rows)")
    ap.add_argument("--config5-share", action="store_true",
                    help="BASELINE config 5, one GPU's share: 125,000 synthetic satellites (seed 20260927) x 10,000 one-minute "
                         "steps, fp32 pos+vel (30 GB); mixed-precision arithmetic unless --f32-arith (packed fp32) / --f32-fp64; parity on sampled rows")
    ap.add_argument("--f32-fp64", action="store_true", help="fp32 outputs from fp64 arithmetic rounded at the store (azh_set_f32_mode(c, 2))")
    ap.add_argument("--f32-arith", action="store_true",
                    help="with fp32 outputs: opt into the packed-fp32-arithmetic kernel (4 m / 6 mm/s) instead of the default "
                         "fp64 arithmetic rounded once at the store (0.25 m / 0.24 mm/s)")
    ap.add_argument("--no-fast-path", action="store_true",
                    help="disable the branch-free uniform-grid step (A/B against the generic tier-voting loop)")
    ap.add_argument("--no-tile-kernel", action="store_true",
                    help="time-major layout: the lane = satellite kernel instead of the 16-satellite tile kernel")
    ap.add_argument("--grid", choices=["uniform", "jdfr", "jitter", "random"], default="uniform",
                    help="time grid of the main workload: exactly uniform one-minute steps (the headline); the (jd, fr) grid of the "
                         "reference's SatrecArray.sgp4 call (uniform to ~4e-7 min); one-minute steps with +-20 s jitter; sorted random "
                         "times (profiling runs: which kernels a grid takes)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default run only: skip the `secondary` block (the other configurations of BASELINE.json, each a "
                         "few steps, timed after the headline region)")
    ap.add_argument("--secondary-skip", default="", help="comma-separated keys of secondary workloads to skip (e.g. config5_share)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    return ap.parse_args()

def main():
    a = parse_args()
    if a.config5_share:
        a.sats, a.times, a.f32_out, a.deep = 125000, 10000, True, 0
        if a.steps == 200 and a.warmup == 100:
            a.steps, a.warmup = 20, 5
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DRY RUN of the N > 1 code path on a box with ONE GPU (tools/gpu_world2_dryrun.sh): every rank on device 0, process group over
    # gloo (RCCL refuses two ranks on one device) -- exercises the barriers, reductions, object gathers, per-rank certificates and
    # the sharded screen with a real world size; its timings mean nothing and the line says so
    dry = os.environ.get("ASTROZ_BENCH_DRYRUN_ONE_DEVICE") == "1"
    if dry:
        local_rank = 0
    if world != a.gpus and world > 1:
        a.gpus = world

    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    torch.cuda.set_device(local_rank)
    if world > 1 or a.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if rank == 0:
        entry.build()          # no-op when the in-tree .so files are current
    if world > 1:
        dist.barrier()         # nobody loads the library before rank 0 has (re)built it
    from astroz_amd import _native, synth
    from astroz_amd.distributed import ShardPlan, ShardedPropagator

    cuda = torch.device("cuda", local_rank)
    n_times = a.times
    times = np.arange(n_times, dtype=np.float64)
    vel_on = not a.pos_only
    odt = torch.float32 if a.f32_out else torch.float64
    sharded = (world > 1 or a.force_sharded) and a.scaling == "strong"      # BASELINE config 4
    gather = sharded and not a.no_gather and not dry      # (gloo has no all-gather of device tensors: the dry run skips the gather)
    mode = {"teme": 0, "ecef": 1, "geodetic": 2}[a.mode]
    ref_jd = 0.0   # set with the workload (synth.START_JD) once the package is imported
    if sharded and (a.layout != "sat" or a.f32_out or mode):
        raise SystemExit("bench.py: the sharded (config 4) path is satellite-major fp64; use --scaling weak for other variants")

    # ---- workload -------------------------------------------------------------------------
    plan = None
    if sharded:
        allp = synth.synth_catalog(n_near=a.sats, n_deep=a.deep, seed=20260926)
        n_total = len(allp)
        # chunk pipeline depth: a window of fewer than ~3,000 rows no longer fills the chip (its launches are latency-bound), so
        # small shards are ONE chunk (8 ranks x 1,685 rows) and only large ones are pipelined against their gathers
        n_chunks = a.chunks if a.chunks > 0 else max(1, min(4, (n_total // world) // 3000))
        plan = ShardPlan(n_total, world, n_chunks)
        pairs = [allp[i] for i in plan.local_rows(rank)]
    else:
        pairs = synth.synth_catalog(n_near=a.sats, n_deep=a.deep, seed=(20260927 if a.config5_share else 20260926) + 101 * rank)
        n_total = len(pairs) * world
    dev = _native.DeviceConstellation.from_tle_lines(pairs, _native.WGS72, local_rank)
    if a.tile:
        dev.set_time_tile(a.tile, a.tile)
    if a.no_fast_path:
        dev.set_fast_path(False)
    if a.no_tile_kernel:
        dev.set_tile_kernel(False)
    dev.set_f32_arithmetic("packed" if a.f32_arith else ("fp64" if a.f32_fp64 else "mixed"))
    n_local = dev.n
    offsets = (synth.START_JD - dev.epochs) * 1440.0
    if a.grid == "jdfr":       # api.py L300-302 on examples/python_sgp4.py's (jd, fr)
        jd_ = np.full(n_times, synth.START_JD)
        fr_ = 0.32853009 + np.arange(n_times) / 1440.0
        rjd_ = jd_[0] + fr_[0]
        times = ((jd_ + fr_) - rjd_) * 1440.0
        offsets = (rjd_ - dev.epochs) * 1440.0
    elif a.grid == "jitter":
        times = times + np.random.default_rng(7).uniform(-1.0 / 3.0, 1.0 / 3.0, n_times)
    elif a.grid == "random":
        times = np.sort(np.random.default_rng(7).uniform(0.0, float(n_times), n_times))
    layout = _native.TIME_MAJOR if a.layout == "time" else _native.SAT_MAJOR
    stride = 0
    if layout == _native.TIME_MAJOR and a.stride_align > 0:
        stride = -(-n_local // a.stride_align) * a.stride_align
    # an explicit (non-null) stream: the kernels and the timing events live on it
    stream = torch.cuda.Stream(device=cuda)
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    assert sptr != 0

    sp = None
    if sharded:
        sp = ShardedPropagator(dev, plan, rank, n_times, velocities=vel_on, device=cuda)
        pos, vel = sp.local[0], (sp.local[1] if vel_on else None)
        p_ptr, v_ptr = pos.data_ptr(), (vel.data_ptr() if vel_on else None)

        def step(do_gather=gather):
            sp.step(gather=do_gather)

        def drain():                      # the launch stream waits for the pipeline's streams
            stream.wait_stream(sp.comm)
            stream.wait_stream(sp.compute)
    else:
        shape = (n_times, stride or n_local, 3) if layout == _native.TIME_MAJOR else (n_local, n_times, 3)
        pos = torch.empty(shape, dtype=odt, device=cuda)
        vel = torch.empty(shape, dtype=odt, device=cuda) if vel_on else None
        p_ptr, v_ptr = pos.data_ptr(), (vel.data_ptr() if vel_on else None)

        def step(do_gather=False):
            dev.propagate_device_cached(p_ptr, v_ptr, layout=layout, stride=stride, stream=sptr, f32=a.f32_out)

        def drain():
            pass
    torch.cuda.synchronize()

    # stage inputs (times, offsets) once; this call also runs the kernels (counts as warm-up)
    ref_jd = synth.START_JD
    dev.propagate_device(times, offsets, p_ptr, v_ptr, mode=mode, reference_jd=ref_jd, layout=layout, stride=stride, stream=sptr,
                         f32=a.f32_out)
    torch.cuda.synchronize()
    last_kernel_ms = dev.last_kernel_ms()   # the library's own hipEvent pair around that launch
    dev.set_timing(False)                    # the timed loop below is bracketed by events of its own
    # Device preconditioning (untimed, reported in the JSON line): measured on MI355X, the clocks move for
    # tens of milliseconds after the onset of a sustained load and then settle.  W warm-up steps of a
    # 0.2-ms kernel end inside that transient unless W is in the hundreds, so the same step is first run back
    # to back for a fixed wall time; the W warm-up steps and the K timed steps follow immediately.
    n_pre = 0
    sampler = PowerSampler(torch, local_rank).start() if rank == 0 else None
    t_pre = time.perf_counter()
    if sharded:
        # the step contains collectives: every rank must run the same number of them
        for _ in range(max(1, int(a.precondition_ms / (5.0 if gather else 0.3)))):
            step()
            n_pre += 1
        drain()
        torch.cuda.synchronize()
    else:
        while (time.perf_counter() - t_pre) * 1e3 < a.precondition_ms:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            n_pre += 20
    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    power = sampler.stop() if sampler else None   # (not sampled during the timed steps: nothing else runs there)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    if sharded:
        sp.compute.wait_stream(stream)
    for _ in range(a.steps):
        step()
    drain()
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if world > 1 or a.force_sharded:
        tt = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=cuda)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(tt[0]), float(tt[1])

    # config 4 extras, timed the same way (barrier, events, max over ranks): the kernels alone -- eager and as hipGraphs -- and
    # the "replicate" point (every GPU propagates the FULL catalog itself): bench_sharded.py
    kernel_only_ms = kernel_graphs_ms = replicate_ms = None
    screen_res = None
    if sharded:
        import bench_sharded
        kernel_only_ms, kernel_graphs_ms = bench_sharded.kernel_only(torch, dist, a, dev, sp, plan, step, drain, stream, cuda)
        replicate_ms = bench_sharded.replicate(torch, dist, _native, synth, a, allp, sp, times, vel_on, stream, sptr, cuda, local_rank)
        try:
            screen_res = bench_sharded.sharded_screen(torch, dist, _native, synth, a, allp, world, rank, times, stream, sptr, cuda, local_rank)
        except Exception as exc:     # (an extra: every rank fails or succeeds alike -- the collectives inside are reached by all or none)
            screen_res = {"failed": repr(exc)}
        if gather:
            # leave the gathered result of the sharded pipeline in sp.full for the parity check below
            sp.compute.wait_stream(stream)
            step()
            drain()
            torch.cuda.synchronize()
            dist.barrier()

    # Every rank certifies its OWN output before anybody leaves ("no gather -- verify by checksums + sampled rows"):
    # sampled rows against the oracle on the rank's host cores, checksum + finiteness over everything it wrote; the certificates
    # travel to rank 0 as one small object.  N = 1 makes the same certificate without a group.
    per_rank = None
    grouped = world > 1 or a.force_sharded
    if (grouped or a.config5_share) and mode == 0:
        try:
            from bench_common import rank_certificate
            if sharded:
                sp.compute.wait_stream(stream)
                step(False)
                drain()
                torch.cuda.synchronize()
            thr_ = max(1, usable_cpus() // max(1, min(world, 8)))
            mine = rank_certificate(torch, pairs, times, offsets, pos, vel, layout == _native.SAT_MAJOR, rank, threads=thr_)
        except Exception as exc:
            mine = {"rank": rank, "failed": repr(exc)}
        per_rank = [mine]
        if grouped:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)

    # Every collective is behind us: all ranks leave the process group HERE, together, and ranks != 0 exit -- rank 0 does the rest
    # (group_host on all devices, oracle parity, the secondary block) alone, with no communicator alive and no peer process
    # holding a GPU in a barrier kernel.
    if world > 1 or a.force_sharded:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    if rank != 0:
        return
    # config 4, host-returning consumers (azh_group_propagate_host: ONE process, all N devices, every device copies its shard
    # into the caller's catalog-ordered host arrays over its own PCIe link, no collective): rank 0 only, behind a watchdog
    group_host, group_host_stuck = None, False
    if sharded:
        import bench_sharded
        group_host, group_host_stuck = bench_sharded.group_host(_native, synth, allp, world, local_rank, times, vel_on, n_total, n_times)

    props_per_step = n_total * n_times
    value = props_per_step * a.steps / elapsed
    launch_s = (ev_ms / 1e3) / a.steps            # average duration of one launch (HIP events on the launch stream)
    if kernel_only_ms is not None:
        launch_s = kernel_only_ms / 1e3           # sharded: the step also holds the collectives
    local_props = n_local * n_times
    bytes_per_launch = local_props * (BYTES_OUT_PV if vel_on else BYTES_OUT_P) * (0.5 if a.f32_out else 1.0) + n_times * 8 + \
        n_local * ELEM_BYTES_PER_SAT  # elements counted once (re-reads across tiles are cache hits)
    gbs = bytes_per_launch / launch_s / 1e9
    tflops = local_props * FLOPS_PER_PROP / launch_s / 1e12

    # HBM traffic per launch from the PMC counters: collected offline with rocprofv3 on this same command
    # (separate --pmc passes, tools/profile.sh) and committed under profiles/ together with a fingerprint of
    # the kernel sources it was measured on; a stale measurement is not reported
    traffic = None
    valu_issue = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        # (the fingerprint was written on the GPU box by tools/profile_run.py when the counters were collected; a file
        # measured on other sources is not reported)
        if (pm.get("csrc_sha16") == csrc_fingerprint() and world == 1 and layout == _native.SAT_MAJOR and vel_on and
                a.sats == 13478 and n_times == 1440 and not a.deep and not a.f32_out and not a.no_fast_path and mode == 0 and
                a.grid == "uniform"):
            pm = pm.get("step", pm)     # tools/profile_run.py keeps the step totals under "step"
            traffic = pm["hbm_bytes_per_launch"]
            if pm.get("valu_wave_insts_per_launch"):
                # VALU issue-slot utilisation of the step: one wave instruction occupies its SIMD's issue port for 4 cycles
                # (16 lanes/cycle); 1,024 SIMDs; shader clock = the median sampled right before the timed steps
                wi = float(pm["valu_wave_insts_per_launch"])
                sclk = (power or {}).get("sclk_mhz_median") or 2400.0
                valu_issue = {"valu_wave_insts_per_launch": wi, "valu_insts_per_propagation": wi * 64.0 / local_props,
                              "sclk_mhz": sclk, "issue_slot_frac": wi * 4.0 / (1024.0 * launch_s * sclk * 1e6),
                              "source": "SQ_INSTS_VALU summed over the step's kernels (rocprofv3 --pmc pass stamped in "
                                        "profiles/latest_pmc.json) x 4 clk / (1,024 SIMDs x avg_launch x shader clock)"}
    except (OSError, ValueError, KeyError):
        pass

    wl, par, arith, kname = describe_workload(a, world, sharded, gather, plan.n_chunks if plan else 0, n_total, n_times, layout == _native.SAT_MAJOR, mode, vel_on)
    out = {
        "metric": ("propagations/sec, 13,478 sats x 1,440 times, at 1/2/4/8 MI355X" if not a.config5_share else
                   "propagations/sec, config 5 (1M sats x 10,000 times, fp32, 8 MI355X): one GPU's 125,000-satellite share"),
        "value": value, "unit": "propagations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
        "scaling": "n/a" if world == 1 else a.scaling, "vs_baseline": None,
        "dtype": "f64" if "fp64 arithmetic" == arith else "f32 (+f64 phase/radius chains)", "data": "synthetic" if not dry else "synthetic; DRY RUN: all ranks on ONE device over gloo -- code-path check, timings meaningless",
        "config": {
            "workload": wl + ("" if a.grid == "uniform" else " [time grid: %s]" % a.grid), "launch_path": dev.last_path(), "n_sats_total": n_total, "n_sats_per_gpu": n_local, "n_times": n_times, "gather": bool(gather),
            "precondition_ms": a.precondition_ms, "precondition_steps": n_pre,
            **({"t_kernel_ms": kernel_only_ms, "t_allgather_ms": max(elapsed / a.steps * 1e3 - kernel_only_ms, 0.0),
                "t_total_ms": elapsed / a.steps * 1e3, "rccl_ranks": world, "chunks": plan.n_chunks,
                "kernel_only_value": props_per_step / (kernel_only_ms / 1e3), "t_kernel_graphs_ms": kernel_graphs_ms,
                "t_replicate_ms": replicate_ms, "replicate_value": props_per_step / (replicate_ms / 1e3),
                "replicate_note": "every GPU propagates the FULL catalog itself (no shards, zero bytes moved): the same "
                                  "deliverable as the gathered run -- the full arrays on every GPU",
                "group_host": group_host, "sharded_screen": screen_res,
                "gather_bytes_per_gpu": (plan.padded - plan.local_capacity()) * n_times * 3 * 8 * (2 if vel_on else 1)}
               if kernel_only_ms is not None else {}),
            "parallelism": par,
        },
        "roofline": {
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": kname,
            "avg_launch_ms": launch_s * 1e3, "first_launch_ms_hipevent": last_kernel_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
        },
        "power": power,
        "valu_issue": valu_issue,
        "fp64_valu": {
            "achieved": tflops, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "flops_per_propagation": FLOPS_PER_PROP,
            "note": "rate x the algorithmic flops of the REFERENCE formulation (405+44K, K=4) -- informational; this kernel "
                    "executes far fewer instructions, see valu_issue for the issue-slot utilisation it actually reaches",
        },
    }

    # ---- parity spot-check + CPU baseline (untimed, rank 0, N=1 only) ------------------------
    if world == 1 and a.config5_share:
        try:
            from oracle import oracle
            rows = np.unique(np.linspace(0, n_local - 1, 24).astype(np.int64))
            cat = oracle.Catalog.from_pairs([pairs[i] for i in rows], oracle.WGS72)
            _, p0, v0 = cat.propagate(times, offsets[rows], layout=oracle.SAT_MAJOR, threads=usable_cpus())
            idx = torch.as_tensor(rows, device=cuda)
            out["parity"] = {"sample_sats": int(len(rows)), "sample": "24 rows spread over the catalog, all 10,000 times, vs the fp64 oracle",
                             "max_dr_km": float(np.linalg.norm(pos[idx].cpu().numpy().astype(np.float64) - p0, axis=2).max())}
            if vel_on:
                out["parity"]["max_dv_kms"] = float(np.linalg.norm(vel[idx].cpu().numpy().astype(np.float64) - v0, axis=2).max())
            out["cpu_baseline"] = {"value": None, "unit": "propagations/s", "cores": 0, "kind": "port",
                                   "sample": "not timed for this flag (see the default run's cpu_baseline)"}
        except Exception as exc:
            out["parity"] = {"failed": repr(exc)}
    elif sharded and gather:
        # config 4: the gathered, catalog-ordered arrays on rank 0 against the oracle on rows spread over every
        # rank's cells (checks the shard placement of the all-gather on the real interconnect)
        try:
            from oracle import oracle
            rows = np.unique(np.linspace(0, n_total - 1, 16 * world * plan.n_chunks).astype(np.int64))
            cat = oracle.Catalog.from_pairs([allp[i] for i in rows], oracle.WGS72)
            offs = (synth.START_JD - cat.epoch_jd) * 1440.0
            _, p0, v0 = cat.propagate(times, offs, layout=oracle.SAT_MAJOR, threads=usable_cpus())
            idx = torch.as_tensor(rows, device=cuda)
            full = sp.results()
            out["parity"] = {"sample_sats": int(len(rows)), "sample": "rows spread over all (chunk, rank) cells of the gathered array on rank 0",
                             "max_abs_dr_km": float(np.abs(full[0][idx].cpu().numpy() - p0).max())}
            if vel_on:
                out["parity"]["max_abs_dv_kms"] = float(np.abs(full[1][idx].cpu().numpy() - v0).max())
        except Exception as exc:
            out["parity"] = {"failed": repr(exc)}
    elif world == 1 and not a.no_cpu_baseline and mode == 0:
        try:
            cb, (n_s, p0, v0) = cpu_baseline(pairs, times, offsets, a.cpu_seconds, layout == _native.SAT_MAJOR)
            out["cpu_baseline"] = cb
            sl = (slice(None), slice(0, n_s)) if layout == _native.TIME_MAJOR else (slice(0, n_s),)
            out["parity"] = {"max_abs_dr_km": float(np.abs(pos[sl].cpu().numpy() - p0).max()), "sample_sats": n_s}
            if vel_on:
                out["parity"]["max_abs_dv_kms"] = float(np.abs(vel[sl].cpu().numpy() - v0).max())
        except Exception as exc:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "propagations/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (exc,)}
    if per_rank is not None:
        out.setdefault("parity", {})
        if isinstance(out["parity"], dict):
            out["parity"]["per_rank"] = per_rank
            out["parity"]["per_rank_note"] = ("every rank: 24 rows spread over ITS catalog x all times vs the fp64 oracle (max_dr km, max_dv km/s), "
                                              "fp64 checksum + finiteness over everything it wrote")
    if (world > 1 or a.force_sharded or a.config5_share) and not a.no_cpu_baseline and mode == 0 and not (out.get("cpu_baseline") or {}).get("value"):
        # N > 1 lines (and the config-5 share) carry a CPU baseline too: rank 0, after leaving the group, on a bounded sample
        try:
            from bench_common import cpu_baseline_sample
            out["cpu_baseline"] = cpu_baseline_sample(pairs, times, offsets, a.cpu_seconds, layout == _native.SAT_MAJOR)
        except Exception as exc:
            out["cpu_baseline"] = {"value": None, "unit": "propagations/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (exc,)}
    if world == 1 and mode != 0 and not a.config5_share and "parity" not in out:
        try:
            from oracle import oracle
            rows = _sample_rows(n_local, 16)
            cat = oracle.Catalog.from_pairs([pairs[i] for i in rows], oracle.WGS72)
            _, p0, v0 = cat.propagate(times, offsets[rows], mode=mode, reference_jd=ref_jd, layout=oracle.SAT_MAJOR, threads=usable_cpus())
            idx = torch.as_tensor(rows, device=cuda)
            gp = (pos[:, idx].permute(1, 0, 2) if layout == _native.TIME_MAJOR else pos[idx]).cpu().numpy().astype(np.float64)
            d = gp - p0
            if mode == 2:
                d[..., 1] = (d[..., 1] + np.pi) % (2 * np.pi) - np.pi
            out["parity"] = {"sample_sats": int(len(rows)), "max_abs_dpos": float(np.abs(d).max()),
                             "units": "km" if mode == 1 else "rad, rad, km"}
            if vel_on:
                gv = (vel[:, idx].permute(1, 0, 2) if layout == _native.TIME_MAJOR else vel[idx]).cpu().numpy().astype(np.float64)
                out["parity"]["max_abs_dv_kms"] = float(np.abs(gv - v0).max())
        except Exception as exc:
            out["parity"] = {"failed": repr(exc)}
    default_workload = (world == 1 and not sharded and not a.config5_share and a.sats == 13478 and n_times == 1440 and not a.deep and
                        vel_on and not a.f32_out and a.layout == "sat" and not a.no_fast_path and not a.tile and mode == 0)
    if default_workload and not a.no_secondary:
        try:
            from bench_secondary import run_secondary
            out["secondary"] = run_secondary(torch, _native, synth, cuda, stream, dev, pairs,
                                             skip=tuple(k for k in a.secondary_skip.split(",") if k))
        except Exception as exc:   # never takes the headline down
            out["secondary"] = [{"failed": repr(exc)}]
    # stdout: the FULL record first (one long line, also written to a file), then -- LAST -- the compact line the driver
    # parses (< 4 KB: bench_common.compact_line).  RCCL writes a version banner through C stdio, which (redirected to a
    # file or a pipe) sits in the C buffer until the process ends -- flush it first
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    full_path = None
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        full_path = os.path.join("gpurun_out", "bench_full.json")
        with open(os.path.join(ROOT, full_path), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        full_path = None
    print("BENCH_FULL " + json.dumps(out), flush=True)
    print(compact_line(out, full_path), flush=True)
    if group_host_stuck:
        os._exit(0)   # (a stuck extra must not keep the process alive)


if __name__ == "__main__":
    main()
