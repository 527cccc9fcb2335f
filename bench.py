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
BYTES_OUT_PV = 48.0          # 6 fp64 written
BYTES_OUT_P = 24.0
ELEM_BYTES_PER_SAT = 32 * 8 + 8 + 4   # element rows read per satellite per time tile + offset + flags
FLOPS_PER_PROP = 581.0       # reference formulation at K = 4 Newton trips (405 + 44 K)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz


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
    ap.add_argument("--chunks", type=int, default=4, help="N > 1: pipeline depth of the compute / all-gather overlap")
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


def usable_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU box
    exposes 256 hardware threads but a 16-CPU quota; more threads than quota only thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(pairs, times, offsets, seconds, sat_major):
    """CPU baseline timed on this host on a bounded sample of the same workload:
      * `value`: oracle/astroz_batch8.c -- the reference's multithreaded SIMD CPU *design* restated in C
        (8 satellites per vector register, polynomial sincos/atan2, all-lane Newton exit, threads over
        batch / time ranges), compiled -O3 -march=native on this host;
      * `scalar_oracle`: the scalar libm oracle (the parity checker), one pass, also used for the
        parity spot check of the GPU output."""
    from oracle import oracle

    olayout = oracle.SAT_MAJOR if sat_major else oracle.TIME_MAJOR
    threads = max(1, min(oracle.max_threads(), usable_cpus()))
    cat = oracle.Catalog.from_pairs(pairs, oracle.WGS72)
    n_s = cat.n
    # scalar oracle: warm the thread pool and the output pages, then one timed pass
    out = cat.propagate(times, offsets[:n_s], layout=olayout, threads=threads)
    t0 = time.perf_counter()
    _, p, v = cat.propagate(times, offsets[:n_s], layout=olayout, threads=threads, out=out)
    scalar_rate = n_s * len(times) / (time.perf_counter() - t0)
    # SIMD-design baseline: whole passes over the catalog until ~`seconds` of wall time
    bout = cat.propagate_batch8(times, offsets[:n_s], layout=olayout, threads=threads)[1:]
    passes, dt = 0, 0.0
    t0 = time.perf_counter()
    while passes == 0 or (dt < seconds and passes < 2000):
        cat.propagate_batch8(times, offsets[:n_s], layout=olayout, threads=threads, out=bout)
        passes += 1
        dt = time.perf_counter() - t0
    # ... and the same code on ONE thread (next to the reference's published 37.7 M/s single-thread figure): whole
    # passes over the first 1,024 satellites for ~1.5 s
    n1 = min(n_s, 1024)
    cat1 = oracle.Catalog.from_pairs(pairs[:n1], oracle.WGS72)
    b1 = cat1.propagate_batch8(times, offsets[:n1], layout=olayout, threads=1)[1:]
    p1, d1 = 0, 0.0
    t0 = time.perf_counter()
    while p1 == 0 or (d1 < 1.5 and p1 < 2000):
        cat1.propagate_batch8(times, offsets[:n1], layout=olayout, threads=1, out=b1)
        p1 += 1
        d1 = time.perf_counter() - t0
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": passes * n_s * len(times) / dt, "unit": "propagations/s", "cores": threads, "kind": "port",
        "threads_1": {"value": p1 * n1 * len(times) / d1, "unit": "propagations/s", "cores": 1,
                      "sample": "%d pass(es) over the first %d satellites x %d times, %.1f s" % (p1, n1, len(times), d1),
                      "reference_published": "37.7 M propagations/s, 1 thread, Ryzen 7 7840U (README.md L35-45 of the reference)"},
        "cpu_model": model, "host_threads_visible": os.cpu_count(),
        "sample": "%d pass(es) over all %d satellites x %d times of the same catalog, %.1f s wall (%.0f core-seconds), "
                  "fp64 pos+vel; C restatement of the reference's SIMD CPU design (8 satellites per AVX-512 register, "
                  "polynomial sincos/atan2, OpenMP over %s ranges), gcc -O3 -march=native" % (
                      passes, n_s, len(times), dt, dt * threads, "batch" if sat_major else "time"),
        "scalar_oracle": {"value": scalar_rate, "unit": "propagations/s", "cores": threads,
                          "note": "scalar libm C oracle (the parity checker), one pass"},
        "reference_published": "303 M propagations/s, 16 threads, Ryzen 7 7840U (README.md of the reference)",
    }, (n_s, p, v)


class PowerSampler:
    """Socket power and shader clock of one GPU while the benchmark loop runs, read from the amdgpu hwmon files (no
    subprocess): the row kernels run on the board's power limit, so the clock they get is part of the result
    (DESIGN.md 4a).  Everything here is best effort: a missing file just leaves its field out."""

    def __init__(self, torch, index, period_s=0.004):
        import glob
        import threading
        self.period, self.samples, self._stop, self.dir = period_s, [], threading.Event(), None
        cands = []
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            cands += glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
        except Exception:
            pass
        if not cands and torch.cuda.device_count() == 1:
            cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        for d in cands:
            if self._read(d, ("power1_average", "power1_input")) is not None:
                self.dir = d
                break
        self.thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(d, names):
        for n in names:
            try:
                with open(os.path.join(d, n)) as f:
                    return float(f.read().strip())
            except (OSError, ValueError):
                continue
        return None

    def _run(self):
        while not self._stop.wait(self.period):
            self.samples.append((self._read(self.dir, ("power1_average", "power1_input")), self._read(self.dir, ("freq1_input",))))

    def start(self):
        if self.dir:
            self.thread.start()
        return self

    def stop(self):
        if not self.dir:
            return None
        self._stop.set()
        self.thread.join(timeout=1.0)
        pw = sorted(p for p, _ in self.samples if p)
        ck = sorted(c for _, c in self.samples if c)
        if not pw:
            return None
        out = {"socket_w_median": pw[len(pw) // 2] / 1e6, "socket_w_max": pw[-1] / 1e6, "samples": len(pw),
               "window": "preconditioning + warm-up steps (the same kernels, immediately before the timed steps)"}
        if ck:
            out["sclk_mhz_median"] = ck[len(ck) // 2] / 1e6
        cap = self._read(self.dir, ("power1_cap",))
        if cap:
            out["socket_w_limit"] = cap / 1e6
        return out


def _sample_rows(n, k):
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))


def run_secondary(torch, _native, synth, cuda, stream, dev2, pairs2, skip=()):
    """The non-headline configurations, each a few steps, run AFTER the headline's timed region on the same device
    and stream (the headline fields never depend on anything here).  One entry per workload: ms_per_step (HIP events
    on the launch stream around K back-to-back steps after W warm-ups), value, HBM roofline fraction of the step, and
    parity against the fp64 oracle on rows spread over the catalog (all times).  A failing entry reports the
    exception and never takes the bench line down."""
    from oracle import oracle
    sptr = stream.cuda_stream
    res = []

    def timed(fn, warm, steps, pre_ms=200.0):
        # the device idles while the CPU baseline / the oracle of the previous entry run: bring the clocks back to their
        # loaded level first (the same back-to-back preconditioning as the headline's, shorter)
        fn()
        torch.cuda.synchronize()
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < pre_ms:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    def case(key, workload, kernel, dev, pairs, n_times, *, layout, vel=True, f32=False, mode=0, steps=50, warm=20,
             cold=False, rows=16, ref_jd=0.0, arith32="mixed", stride_align=0, grid="uniform"):
        if key in skip:
            return
        ent = {"key": key, "workload": workload, "kernel": kernel}
        try:
            n = dev.n
            dev.set_f32_arithmetic(arith32)
            times = np.arange(n_times, dtype=np.float64)
            offs = (synth.START_JD - dev.epochs) * 1440.0
            if grid == "jdfr":
                # the reference's own call: SatrecArray.sgp4(jd, fr) with jd = full(n, day), fr = f0 + arange(n)/1440
                # (examples/python_sgp4.py L31-33) -> times = ((jd + fr) - reference_jd) * 1440 (api.py L300-302): uniform only to
                # ~4e-7 min after the rounding of jd + fr at 2.46e6 days
                jd = np.full(n_times, synth.START_JD)
                fr = 0.32853009 + np.arange(n_times) / 1440.0
                rjd = jd[0] + fr[0]
                times = ((jd + fr) - rjd) * 1440.0
                offs = (rjd - dev.epochs) * 1440.0
                st_ = (times[-1] - times[0]) / (n_times - 1)
                ent["grid"] = {"kind": "jd+fr (api.py L300-302)", "max_dev_from_uniform_min": float(np.abs(times - (times[0] + np.arange(n_times) * st_)).max())}
            elif grid == "jitter":
                # one-minute grid with +-20 s of jitter per point: "uniform with jitter" (fast_step.h, the wide DELTA form)
                times = times + np.random.default_rng(7).uniform(-1.0 / 3.0, 1.0 / 3.0, n_times)
                ent["grid"] = {"kind": "one-minute steps + uniform(-20 s, 20 s) jitter"}
            elif grid == "random":
                # sorted random times over the same day: no uniform structure at all, the generic kernels
                times = np.sort(np.random.default_rng(7).uniform(0.0, float(n_times), n_times))
                ent["grid"] = {"kind": "sorted uniform-random times over the span"}
            stride = (n + stride_align - 1) // stride_align * stride_align if (stride_align and layout == _native.TIME_MAJOR) else n
            shape = (n_times, stride, 3) if layout == _native.TIME_MAJOR else (n, n_times, 3)
            odt = torch.float32 if f32 else torch.float64
            pos = torch.empty(shape, dtype=odt, device=cuda)
            v = torch.empty(shape, dtype=odt, device=cuda) if vel else None
            pp, vp = pos.data_ptr(), (v.data_ptr() if vel else None)
            dev.propagate_device(times, offs, pp, vp, mode=mode, reference_jd=ref_jd, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32)
            torch.cuda.synchronize()
            ent["path"] = dev.last_path()  # azh_last_path: 1 k_rows_fast, 2 k_tiles_fast, 4 k_rows, 8 k_propagate, 16 k_rows_deep, 32 quasi-uniform form
            ms = timed(lambda: dev.propagate_device_cached(pp, vp, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32), warm, steps)
            props = n * n_times
            nbytes = props * (BYTES_OUT_PV if vel else BYTES_OUT_P) * (0.5 if f32 else 1.0) + n_times * 8 + n * ELEM_BYTES_PER_SAT
            ent.update({"ms_per_step": ms, "value": props / (ms / 1e3), "unit": "propagations/s", "steps": steps, "warmup": warm,
                        "n_sats": n, "n_times": n_times, "dtype_out": "f32" if f32 else "f64",
                        "roofline": {"bound": "hbm", "achieved": nbytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}})
            if cold:
                # a NEW time grid every call: input staging (H2D of times/offsets), k_prep_inc, k_deep_seed and the step
                # itself, host wall clock around call + synchronize
                cs = []
                for j in range(5):
                    tj = times + 0.25 * (j + 1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dev.propagate_device(tj, offs, pp, vp, mode=mode, reference_jd=ref_jd, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32)
                    torch.cuda.synchronize()
                    cs.append((time.perf_counter() - t0) * 1e3)
                ent["cold_grid_call_ms"] = {"median": sorted(cs)[len(cs) // 2], "min": min(cs),
                                            "what": "first call on a new time grid: H2D staging + k_prep_inc + k_deep_seed + the step, "
                                                    "host wall clock incl. the final synchronize"}
                dev.propagate_device(times, offs, pp, vp, mode=mode, reference_jd=ref_jd, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32)
                torch.cuda.synchronize()
            # parity on sampled rows, all times
            rws = _sample_rows(n, rows)
            cat = oracle.Catalog.from_pairs([pairs[i] for i in rws], oracle.WGS72)
            _, p0, v0 = cat.propagate(times, offs[rws], mode=mode, reference_jd=ref_jd, layout=oracle.SAT_MAJOR,
                                      threads=usable_cpus())
            idx = torch.as_tensor(rws, device=cuda)
            take = (lambda x: x[:, idx].permute(1, 0, 2)) if layout == _native.TIME_MAJOR else (lambda x: x[idx])
            gp = take(pos).cpu().numpy().astype(np.float64)
            if mode == 2:
                # geodetic rows are (lat rad, lon rad, alt km): longitude differences modulo 2 pi
                d = gp - p0
                d[..., 1] = (d[..., 1] + np.pi) % (2 * np.pi) - np.pi
                ent["parity"] = {"rows": int(len(rws)), "max_abs_dlatlon_rad": float(np.abs(d[..., :2]).max()),
                                 "max_abs_dalt_km": float(np.abs(d[..., 2]).max())}
            else:
                ent["parity"] = {"rows": int(len(rws)), "max_abs_dr_km": float(np.abs(gp - p0).max())}
            if vel:
                ent["parity"]["max_abs_dv_kms"] = float(np.abs(take(v).cpu().numpy().astype(np.float64) - v0).max())
            del pos, v
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)

    TM, SM = _native.TIME_MAJOR, _native.SAT_MAJOR
    ref_jd = synth.START_JD
    n2 = dev2.n
    case("config2_pos_only", "config 2, positions only (%d x 1,440, fp64 TEME, satellite-major)" % n2,
         "k_rows_fast<pos> + redo", dev2, pairs2, 1440, layout=SM, vel=False)
    case("config2_time_major", "config 2, TIME-major output (the reference benchmark's physical layout, api.py L304-314), fp64 TEME pos+vel",
         "k_tiles_fast<pos+vel> + redo", dev2, pairs2, 1440, layout=TM)
    case("config2_time_major_aligned", "config 2, TIME-major output with the time rows padded to a multiple of 16 satellites (out_stride_sats = "
         "13,488: every 384-byte tile run on whole cache lines; what SatrecArray.sgp4_device(padded=True) allocates), fp64 TEME pos+vel",
         "k_tiles_fast<pos+vel> (streaming flush) + redo", dev2, pairs2, 1440, layout=TM, stride_align=16)
    case("config2_time_major_jdfr", "config 2, TIME-major, on the grid the reference's own API call produces: SatrecArray.sgp4(jd, fr), "
         "times = ((jd + fr) - reference_jd) * 1440 (quasi-uniform: first-order correction of every point to its rounded time)",
         "k_tiles_fast<pos+vel,DELTA> + redo", dev2, pairs2, 1440, layout=TM, grid="jdfr")
    case("config2_sat_major_jdfr", "config 2, satellite-major, on the (jd, fr) grid of the reference's API call",
         "k_rows_fast<pos+vel,DELTA> + redo", dev2, pairs2, 1440, layout=SM, grid="jdfr")
    case("config2_time_major_irregular", "config 2, TIME-major, one-minute steps with +-20 s jitter (VERDICT r03's irregular grid: uniform with jitter, "
         "the wide quasi-uniform form)", "k_tiles_fast<pos+vel,DELTA=2> + redo", dev2, pairs2, 1440, layout=TM, grid="jitter", steps=20, warm=5)
    case("config2_sat_major_irregular", "config 2, satellite-major, one-minute steps with +-20 s jitter",
         "k_rows_fast<pos+vel,DELTA=2> + redo", dev2, pairs2, 1440, layout=SM, grid="jitter", steps=20, warm=5)
    case("config2_time_major_random", "config 2, TIME-major, sorted random times (no uniform structure: the generic kernels)",
         "k_propagate<time-major> (lane = satellite, generic step)", dev2, pairs2, 1440, layout=TM, grid="random", steps=20, warm=5)
    case("config2_sat_major_random", "config 2, satellite-major, sorted random times",
         "k_rows (generic, lane = time)", dev2, pairs2, 1440, layout=SM, grid="random", steps=20, warm=5)
    case("config2_ecef_time_major", "config 2, ECEF time-major (the default of the reference's high-level propagate(), "
         "Constellation.zig L489-506), fp64 pos+vel", "k_tiles_fast<pos+vel,ECEF> + redo", dev2, pairs2, 1440, layout=TM, mode=1, ref_jd=ref_jd)
    case("config2_ecef_sat_major", "config 2, ECEF satellite-major, fp64 pos+vel", "k_rows_fast<pos+vel,FRAME> + redo",
         dev2, pairs2, 1440, layout=SM, mode=1, ref_jd=ref_jd)
    case("config2_geodetic_time_major", "config 2, geodetic (lat, lon [rad], alt km) time-major, positions only",
         "k_propagate<time-major,pos,FRAME> (lane = satellite)", dev2, pairs2, 1440, layout=TM, vel=False, mode=2, ref_jd=ref_jd, steps=10)
    if "config1" not in skip:
        ent = {"key": "config1", "kernel": "k_one_satellite (the kernel reads the times from and writes into a pinned buffer itself) / "
                                            "the constellation kernels on a 1-satellite catalog",
               "workload": "BASELINE config 1: the ISS TLE x 1,440 one-minute steps through the Python API mirror (examples/python_sgp4.py "
                           "L31-33): Satrec.sgp4_array(jd, fr), SatrecArray([sat]).sgp4(jd, fr) and 1,440 scalar Satrec.sgp4 calls; host "
                           "wall clock per call, host arrays in and out (reference: 30.8 M/s single-thread sgp4_array, README.md L25-33)"}
        try:
            from astroz_amd.api import Satrec, SatrecArray, WGS72
            l1 = "1 25544U 98067A   24127.82853009  .00015698  00000+0  27310-3 0  9995"
            l2 = "2 25544  51.6393 160.4574 0003580 140.6673 205.7250 15.50957674452123"
            sat = Satrec.twoline2rv(l1, l2, WGS72)
            jd = np.full(1440, sat.jdsatepoch)
            fr = sat.jdsatepochF + np.arange(1440) / 1440.0

            def wall(fn, k):
                for _ in range(max(3, k // 10)):
                    fn()
                t0 = time.perf_counter()
                for _ in range(k):
                    fn()
                return (time.perf_counter() - t0) / k * 1e6
            us_arr = wall(lambda: sat.sgp4_array(jd, fr), 300)
            sa = SatrecArray([sat], device=cuda.index or 0)
            us_sa = wall(lambda: sa.sgp4(jd, fr), 200)
            us_one = wall(lambda: sat.sgp4(jd[0], fr[7]), 500)
            e_, r_, v_ = sat.sgp4_array(jd, fr)
            e2, r2, v2 = sa.sgp4(jd, fr)
            cat = oracle.Catalog.from_pairs([(l1, l2)], oracle.WGS72)
            ts = ((jd + fr) - (sat.jdsatepoch + sat.jdsatepochF)) * 1440.0
            _, p0, v0 = cat.propagate(ts, None, layout=oracle.SAT_MAJOR)
            ent.update({"ms_per_step": us_arr / 1e3, "value": 1440 / (us_arr / 1e6), "unit": "propagations/s (Satrec.sgp4_array, host arrays)",
                        "sgp4_array_us": us_arr, "satrec_array_sgp4_us": us_sa, "scalar_sgp4_us": us_one,
                        "parity": {"max_abs_dr_km": float(max(np.abs(r_ - p0[0]).max(), np.abs(r2[0] - p0[0]).max())),
                                   "max_abs_dv_kms": float(max(np.abs(v_ - v0[0]).max(), np.abs(v2[0] - v0[0]).max())),
                                   "err_nonzero": int(np.count_nonzero(e_) + np.count_nonzero(e2))}})
            del sa
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "api_host" not in skip:
        ent = {"key": "api_host", "kernel": "k_tiles_fast<pos+vel,DELTA> + redo, then device -> host over PCIe",
               "workload": "the reference's flagship Python call, host arrays out: SatrecArray.sgp4(jd, fr) -> (e, r, v) numpy, %d x 1,440, "
                           "fp64 TEME pos+vel, time-major physical layout (api.py L296-320; the call the reference's 290 M/s figure is "
                           "quoted on).  Host wall clock per call INCLUDING fresh result arrays, staging, kernels and the two 466-MB "
                           "device-to-host copies" % n2}
        try:
            from astroz_amd.api import Satrec, SatrecArray
            arr = SatrecArray([Satrec.twoline2rv(a_, b_) for a_, b_ in pairs2], device=cuda.index or 0)
            jd = np.full(1440, synth.START_JD)
            fr = 0.32853009 + np.arange(1440) / 1440.0

            def calls(k):
                ws = []
                e_ = r_ = v_ = None
                for _ in range(k):
                    del e_, r_, v_          # (the previous result is returned to the OS outside the timed call)
                    t0 = time.perf_counter()
                    res_ = arr.sgp4(jd, fr)
                    ws.append((time.perf_counter() - t0) * 1e3)
                    e_, r_, v_ = res_
                    del res_
                return ws, (e_, r_, v_)
            calls(2)
            ws, (e_, r_, v_) = calls(7)
            _native.set_host_copy_threads(0)           # the plain path: pageable D2H straight into the fresh arrays
            ws0, _ = calls(3)
            _native.set_host_copy_threads(-1)
            ms = sorted(ws)[len(ws) // 2]
            out_bytes = r_.nbytes + v_.nbytes + e_.nbytes
            ent.update({"ms_per_step": ms, "min_ms": min(ws), "value": n2 * 1440 / (ms / 1e3), "unit": "propagations/s (host arrays, PCIe-inclusive)",
                        "n_sats": n2, "n_times": 1440, "calls_ms": ws, "d2h_GB_per_s_of_wall": out_bytes / (ms / 1e3) / 1e9,
                        "path": arr._dev.last_path(),
                        "direct_pageable_copy_ms": sorted(ws0)[len(ws0) // 2],
                        "what": "results travel device -> pinned staging slots -> the fresh numpy arrays, the second hop by host threads while "
                                "the next chunk is on the link (azh_set_host_copy_threads); direct_pageable_copy_ms: the same call with "
                                "plain pageable D2H copies (the runtime pins the fresh range first).  PCIe Gen5 x16 moves the 932 MB in "
                                "~16.4 ms: that, not the 0.3-ms kernel, bounds this call"})
            rws = _sample_rows(n2, 16)
            cat = oracle.Catalog.from_pairs([pairs2[i] for i in rws], oracle.WGS72)
            rjd = jd[0] + fr[0]
            _, p0, v0 = cat.propagate(((jd + fr) - rjd) * 1440.0, (rjd - arr._epochs[rws]) * 1440.0, layout=oracle.SAT_MAJOR)
            ent["parity"] = {"rows": int(len(rws)), "max_abs_dr_km": float(np.abs(r_[rws] - p0).max()),
                             "max_abs_dv_kms": float(np.abs(v_[rws] - v0).max()), "err_nonzero": int(np.count_nonzero(e_))}
            del arr, e_, r_, v_
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    dev3 = pairs3 = None
    if not {"config3_sat_major", "config3_time_major"} <= set(skip):
        pairs3 = synth.synth_catalog(n_near=13478, n_deep=1522, seed=20260926)
        dev3 = _native.DeviceConstellation.from_tle_lines(pairs3, _native.WGS72, cuda.index or 0)
        dev3.set_timing(False)
    if dev3 is not None:
        case("config3_sat_major", "config 3: 13,478 near-earth + 1,522 deep-space SDP4 x 1,440, fp64 TEME pos+vel, satellite-major "
             "(steady: resonance seeds of the grid cached in the handle; cold_grid_call_ms: a new grid every call)",
             "k_rows_fast + k_rows_deep + redo (+ k_deep_seed, k_prep_inc on a new grid)", dev3, pairs3, 1440, layout=SM, cold=True, rows=48)
        case("config3_time_major", "config 3, TIME-major output, fp64 TEME pos+vel",
             "k_tiles_fast + deep-space rows + redo", dev3, pairs3, 1440, layout=TM, rows=48)
        dev3.close()
    if "fused_screen" not in skip:
        ent = {"key": "fused_screen", "kernel": "k_rows_fast<SINK = screen> (+ generic pass over rejected windows), k_screen_finalize",
               "workload": "config 2 catalog, fused propagate + single-target conjunction screen (src/Constellation.zig L683-756): minimum "
                           "distance and its grid point of every satellite against satellite 0 over 1,440 steps, nothing stored; one "
                           "call = input staging + window plan + the kernels (azh_screen_target_host)"}
        try:
            times = np.arange(1440, dtype=np.float64)
            offs = (synth.START_JD - dev2.epochs) * 1440.0
            ks, ws = [], []
            dev2.set_timing(True)          # (the library's own event pair around the screen's kernels)
            for _ in range(8):
                t0 = time.perf_counter()
                d, ti = dev2.screen_target(times, 0, 500.0, offs)
                ws.append((time.perf_counter() - t0) * 1e3)
                ks.append(dev2.last_kernel_ms())
            dev2.set_timing(False)
            ms = sorted(ks)[len(ks) // 2]
            ent.update({"ms_per_step": ms, "value": dev2.n * 1440 / (ms / 1e3), "unit": "propagations/s",
                        "call_wall_ms_median": sorted(ws)[len(ws) // 2],
                        "what": "ms_per_step: the library's HIP event pair around the screen's kernels; call_wall_ms: host wall clock of the "
                                "whole call incl. staging and the D2H of the two result vectors"})
            cat = oracle.Catalog.from_pairs(pairs2, oracle.WGS72)
            d0, t0_ = cat.screen_target(times, 0, 500.0, offs)
            ent["parity"] = {"rows": int(dev2.n), "max_abs_dmin_km": float(np.abs(d - d0).max()),
                             "t_index_mismatches": int((ti != t0_).sum())}
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "screen_all" not in skip:
        ent = {"key": "screen_all", "kernel": "k_tiles_fast / k_propagate + the cell-list screen kernels (azh_screen_all_host)",
               "workload": "config 2 catalog, all-vs-all conjunction screen (SURVEY 8 f3; bindings/python/astroz/__init__.py L535-658, "
                           "bindings/python/src/conjunction.zig L152-260): every pair closer than 10 km at any of 120 one-minute "
                           "steps, propagate + spatial hash on the GPU, the positions never leave HBM; host wall clock per call"}
        try:
            times = np.arange(120, dtype=np.float64)
            offs = (synth.START_JD - dev2.epochs) * 1440.0
            ws = []
            for _ in range(4):
                t0 = time.perf_counter()
                pr, tt = dev2.screen_all(times, 10.0, offs)
                ws.append((time.perf_counter() - t0) * 1e3)
            ms = sorted(ws[1:])[len(ws[1:]) // 2]
            ent.update({"ms_per_step": ms, "calls_ms": ws, "value": dev2.n * len(times) / (ms / 1e3), "unit": "propagations/s screened",
                        "pairs_found": int(len(tt))})
            # parity: the same screen by the oracle on the first 24 steps (propagate + its own cell list), as sets of (t, i, j)
            cat = oracle.Catalog.from_pairs(pairs2, oracle.WGS72)
            _, rp, _ = cat.propagate(times[:24], offs, layout=oracle.SAT_MAJOR, velocities=False)
            pr0, tt0 = oracle.coarse_screen(rp, 10.0)
            keep = np.asarray(tt) < 24
            got = set(zip(np.asarray(tt)[keep].tolist(), np.asarray(pr)[keep, 0].tolist(), np.asarray(pr)[keep, 1].tolist()))
            want = set(zip(np.asarray(tt0).tolist(), np.asarray(pr0)[:, 0].tolist(), np.asarray(pr0)[:, 1].tolist()))
            ent["parity"] = {"steps": 24, "pairs": len(want), "missing": len(want - got), "extra": len(got - want)}
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "one_satellite" not in skip:
        ent = {"key": "one_satellite", "kernel": "k_one_fast (every wave fits its own 1,024 points; k_one_satellite behind it for what it hands over)",
               "workload": "one satellite (ISS-like, near-earth) x 10,000,000 times through azh_propagate_one_device: device-resident "
                           "tsince in, pos+vel out (reference: 30.8 M/s single-thread sgp4_array, README.md L25-33)"}
        try:
            n = 10_000_000
            ts = torch.linspace(0.0, 14400.0, n, dtype=torch.float64, device=cuda)
            po = torch.empty((n, 3), dtype=torch.float64, device=cuda)
            ve = torch.empty((n, 3), dtype=torch.float64, device=cuda)
            ms = timed(lambda: dev2.propagate_one_device(0, ts.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sptr), 2, 5)
            nbytes = n * 56.0
            ent["segments_fast_handed_over"] = list(dev2.last_one_stats())
            # the same count of sorted RANDOM times: every segment is handed over to the generic kernel (one full step per point)
            tr = torch.sort(torch.rand(n, dtype=torch.float64, device=cuda, generator=torch.Generator(device=cuda).manual_seed(3)) * 14400.0).values
            ms_irr = timed(lambda: dev2.propagate_one_device(0, tr.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sptr), 1, 3)
            ent["irregular_times"] = {"ms_per_step": ms_irr, "value": n / (ms_irr / 1e3), "segments_fast_handed_over": list(dev2.last_one_stats()),
                                      "frac": nbytes / (ms_irr / 1e3) / 1e9 / HBM_PEAK_GBS}
            dev2.propagate_one_device(0, ts.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sptr)   # (po / ve: the uniform series again, for the parity below)
            torch.cuda.synchronize()
            del tr
            ent.update({"ms_per_step": ms, "value": n / (ms / 1e3), "unit": "propagations/s",
                        "roofline": {"bound": "hbm", "achieved": nbytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}})
            cat = oracle.Catalog.from_pairs([pairs2[0]], oracle.WGS72)
            pick = _sample_rows(n, 4096)
            tsel = ts[torch.as_tensor(pick, device=cuda)].cpu().numpy()
            _, p0, v0 = cat.propagate(tsel, None, layout=oracle.SAT_MAJOR)
            ent["parity"] = {"points": int(len(pick)),
                             "max_abs_dr_km": float(np.abs(po[torch.as_tensor(pick, device=cuda)].cpu().numpy() - p0[0]).max()),
                             "max_abs_dv_kms": float(np.abs(ve[torch.as_tensor(pick, device=cuda)].cpu().numpy() - v0[0]).max())}
            # the same series through HOST pointers (Satrec.sgp4_array / sgp4_propagate_batch: tsince in, e / r / v numpy out)
            th = np.linspace(0.0, 14400.0, n)
            hw = []
            for _ in range(4):
                t0 = time.perf_counter()
                eh, rh, vh = dev2.propagate_one(0, th)
                hw.append((time.perf_counter() - t0) * 1e3)
                del eh, rh, vh
            ent["host_pointers"] = {"ms_per_call": sorted(hw[1:])[len(hw[1:]) // 2], "calls_ms": hw,
                                    "value": n / (sorted(hw[1:])[len(hw[1:]) // 2] / 1e3), "unit": "propagations/s (80 MB in, 490 MB out over PCIe, fresh arrays)"}
            del ts, po, ve, th
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "ingest" not in skip:
        ent = {"key": "ingest", "kernel": "host text reader (azh::parse_all, threads) + k_init",
               "workload": "SURVEY 8-f4: 13,478 TLEs of text -> device-resident constellation (azh_constellation_from_tle_text: parse, "
                           "H2D, element initialisation kernel, classification), host wall clock; and the host reader alone on the "
                           "same text repeated to 1,010,850 records (140 MB, config 5's catalog size)"}
        try:
            text = "\n".join(a + "\n" + b for a, b in pairs2).encode()
            n2 = len(pairs2)
            ws = []
            for _ in range(5):
                t0 = time.perf_counter()
                d = _native.DeviceConstellation.from_tle_text(text, _native.WGS72, cuda.index or 0)
                d.synchronize()
                ws.append((time.perf_counter() - t0) * 1e3)
                ok_n = d.n
                d.close()
            big = text + b"\n"
            big = big * 75
            cap = len(big) // 138 + 1
            buf = np.empty((cap, 16))
            k = ctypes.c_size_t(0)
            L = _native.lib()
            per = {}
            for thr in (1, 0):
                _native.set_parse_threads(thr)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    L.azh_parse_tle_text(big, len(big), buf.ctypes.data, cap, ctypes.byref(k))
                    ts.append((time.perf_counter() - t0) * 1e3)
                per["serial" if thr == 1 else "threads"] = {"ms": min(ts), "records": int(k.value), "records_per_s": k.value / (min(ts) / 1e3)}
            _native.set_parse_threads(0)
            t0 = time.perf_counter()
            d = _native.DeviceConstellation.from_tle_text(big, _native.WGS72, cuda.index or 0)
            d.synchronize()
            per["text_to_device"] = {"ms": (time.perf_counter() - t0) * 1e3, "records": int(d.n)}
            d.close()
            ent.update({"ms_per_step": float(np.median(ws)), "value": n2 / (float(np.median(ws)) / 1e3), "unit": "TLEs/s (text -> initialised on device)",
                        "n_sats": int(ok_n), "min_ms": float(min(ws)), "reader_1M": per, "host_cores": os.cpu_count()})
            del big, buf
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if not {"config5_share", "config5_share_f32arith", "config5_share_fp64"} <= set(skip):
        try:
            pairs5 = synth.synth_catalog(n_near=125000, n_deep=0, seed=20260927)
            dev5 = _native.DeviceConstellation.from_tle_lines(pairs5, _native.WGS72, cuda.index or 0)
            dev5.set_timing(False)
            c5 = "config 5, ONE GPU's share of 8: 125,000 synthetic satellites (seed 20260927) x 10,000 one-minute steps, fp32 pos+vel (30 GB), satellite-major, "
            case("config5_share", c5 + "DEFAULT arithmetic: mixed precision (O(1) quantities fp64, small ones packed fp32; eccentric members fp64 rounded at the store)",
                 "k_rows_fast32<MIXED> (+ eccentric members, redo)", dev5, pairs5, 10000, layout=SM, f32=True, steps=5, warm=2, rows=24)
            case("config5_share_fp64", c5 + "fp64 arithmetic, every component rounded once at the store (azh_set_f32_mode(c, 2))",
                 "k_rows_fast<SINK_F32> (+ eccentric members, redo)", dev5, pairs5, 10000, layout=SM, f32=True, steps=5, warm=2, rows=24,
                 arith32="fp64")
            case("config5_share_f32arith", c5 + "OPT-IN packed fp32 arithmetic (azh_set_f32_mode(c, 1): 4 m / 6 mm/s)",
                 "k_rows_fast32 (+ eccentric members, redo)", dev5, pairs5, 10000, layout=SM, f32=True, steps=5, warm=2, rows=24,
                 arith32="packed")
            dev5.close()
        except Exception as exc:
            res.append({"key": "config5_share", "failed": repr(exc)})
    # the entries for the reference's own call shapes go LAST (a record that keeps only the tail of this line keeps them)
    last = ["config2_time_major", "config2_sat_major_jdfr", "config2_time_major_jdfr", "api_host"]
    res.sort(key=lambda e: last.index(e["key"]) if e.get("key") in last else -1)
    return res


def csrc_fingerprint():
    """sha256 (16 hex digits) over the kernel sources: ties a committed PMC measurement to the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "astroz_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    a = parse_args()
    if a.config5_share:
        a.sats, a.times, a.f32_out, a.deep = 125000, 10000, True, 0
        if a.steps == 200 and a.warmup == 100:
            a.steps, a.warmup = 20, 5
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    torch.cuda.set_device(local_rank)
    if world > 1 or a.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
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
    gather = sharded and not a.no_gather
    mode = {"teme": 0, "ecef": 1, "geodetic": 2}[a.mode]
    ref_jd = 0.0   # set with the workload (synth.START_JD) once the package is imported
    if sharded and (a.layout != "sat" or a.f32_out or mode):
        raise SystemExit("bench.py: the sharded (config 4) path is satellite-major fp64; use --scaling weak for other variants")

    # ---- workload -------------------------------------------------------------------------
    plan = None
    if sharded:
        allp = synth.synth_catalog(n_near=a.sats, n_deep=a.deep, seed=20260926)
        n_total = len(allp)
        plan = ShardPlan(n_total, world, a.chunks)
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

    # config 4: the kernels alone, timed the same way, so that t_kernel and t_allgather are reported next to
    # t_total
    kernel_only_ms = None
    if sharded:
        dist.barrier()
        torch.cuda.synchronize()
        k0 = torch.cuda.Event(enable_timing=True)
        k1 = torch.cuda.Event(enable_timing=True)
        k0.record(stream)
        sp.compute.wait_stream(stream)
        for _ in range(a.steps):
            step(False)
        drain()
        k1.record(stream)
        torch.cuda.synchronize()
        kt = torch.tensor([k0.elapsed_time(k1) / a.steps], dtype=torch.float64, device=cuda)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        kernel_only_ms = float(kt[0])

    # config 4, the alternative DESIGN.md 6 recommends to consumers that need everything everywhere: every GPU propagates
    # the FULL catalog itself ("replicate": zero bytes moved), timed the same way (barrier, events, max over ranks)
    replicate_ms = None
    if sharded:
        dev_full = _native.DeviceConstellation.from_tle_lines(allp, _native.WGS72, local_rank)
        dev_full.set_timing(False)
        offs_full = (synth.START_JD - dev_full.epochs) * 1440.0
        fp, fv = sp.full[0].data_ptr(), (sp.full[1].data_ptr() if vel_on else None)   # (padded >= n_total) x n_times x 3
        dev_full.propagate_device(times, offs_full, fp, fv, layout=_native.SAT_MAJOR, stream=sptr)
        for _ in range(max(5, a.warmup // 4)):
            dev_full.propagate_device_cached(fp, fv, layout=_native.SAT_MAJOR, stream=sptr)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        r0 = torch.cuda.Event(enable_timing=True)
        r1 = torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(a.steps):
            dev_full.propagate_device_cached(fp, fv, layout=_native.SAT_MAJOR, stream=sptr)
        r1.record(stream)
        torch.cuda.synchronize()
        rt = torch.tensor([r0.elapsed_time(r1) / a.steps], dtype=torch.float64, device=cuda)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        replicate_ms = float(rt[0])
        if gather:
            # leave the gathered result of the sharded pipeline in sp.full for the parity check below
            sp.compute.wait_stream(stream)
            step()
            drain()
            torch.cuda.synchronize()
            dist.barrier()

    # config 4, host-returning consumers (the C-host route, azh_group_propagate_host): ONE process, all N devices, every
    # device copies its shard straight into the caller's catalog-ordered host arrays over its own PCIe link -- no collective.
    # The one case where sharding this path pays: the call is PCIe-bound (16.4 ms for 932 MB over one link).  Rank 0 only,
    # after the timed region and after the other ranks have gone.
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
    group_host = None
    if sharded:
        box = {}

        def _group_host():
            try:
                text = "\n".join(x + "\n" + y for x, y in allp)
                devs = list(range(world)) if world > 1 else [local_rank]
                grp = _native.DeviceGroup(text, devs, _native.WGS72, n_chunks=1)
                goff = (synth.START_JD - grp.epochs) * 1440.0
                ws = []
                gp = gv = None
                for _ in range(5):
                    del gp, gv              # (the previous result is returned to the OS outside the timed call)
                    t0 = time.perf_counter()
                    res_ = grp.propagate_host(times, goff, velocities=vel_on)
                    ws.append((time.perf_counter() - t0) * 1e3)
                    gp, gv = res_[0], res_[1]
                    del res_
                gms = sorted(ws[1:])[len(ws[1:]) // 2]
                box["r"] = {"ms_per_call": gms, "calls_ms": ws, "devices": len(devs), "value": n_total * n_times / (gms / 1e3),
                            "GB_per_s": (gp.nbytes + (gv.nbytes if gv is not None else 0)) / (gms / 1e3) / 1e9,
                            "what": "azh_group_propagate_host: one process, N devices, fresh host arrays each call; wall clock"}
                del gp, gv
                grp.close()
            except Exception as exc:
                box["r"] = {"failed": repr(exc)}

        # (a watchdog: this is an extra, it must never take the headline line down with it)
        import threading
        th = threading.Thread(target=_group_host, daemon=True)
        th.start()
        th.join(timeout=120.0)
        group_host = box.get("r", {"failed": "timed out after 120 s"})
        group_host_stuck = th.is_alive()
    else:
        group_host_stuck = False

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

    if world == 1 and a.config5_share:
        wl = "config 5, ONE GPU's share of 8: %d synthetic satellites (seed 20260927) x %d one-minute steps" % (a.sats, n_times)
        par = "single GPU (1/8 of the 1M-satellite job; shards are independent, no gather)"
    elif world == 1:
        wl = ("config 3: %d-sat synthetic catalog (%d SGP4 near-earth + %d deep-space SDP4) x %d one-minute steps" % (
            a.sats + a.deep, a.sats, a.deep, n_times)) if a.deep else (
            "config 2: %d-sat synthetic active catalog (SGP4 near-earth) x %d one-minute steps" % (a.sats, n_times))
        par = "single GPU"
    elif sharded:
        wl = "config 4: the %d-sat synthetic catalog%s x %d one-minute steps, block-cyclic satellite shards over %d GPUs%s" % (
            n_total, " (incl. %d deep-space SDP4)" % a.deep if a.deep else "", n_times, world,
            ", RCCL all-gather of the full result onto every GPU (%d-chunk compute/gather pipeline)" % plan.n_chunks
            if gather else ", NO gather (--no-gather: every GPU keeps its shard)")
        par = "satellite-sharded x%d%s" % (world, " + RCCL all-gather" if gather else ", no data-path collective")
    else:
        wl = "WEAK scaling (--scaling weak): %d GPUs x an own %d-sat synthetic catalog x %d one-minute steps, no gather" % (
            world, a.sats + a.deep, n_times)
        par = "independent catalogs x%d, no data-path collective" % world
    f32_fast = a.f32_out and not a.f32_fp64 and not a.no_fast_path and layout == _native.SAT_MAJOR and mode == 0
    arith = "fp64 arithmetic" if not f32_fast else (
        "packed fp32 arithmetic with fp64 phase and radius chains (near-circular members; fp64 for the rest)" if a.f32_arith else
        "mixed-precision arithmetic (O(1) quantities fp64, small ones packed fp32; near-circular members; fp64 for the rest)")
    wl += ", %s, %s %s %s, %s-major device-resident output" % (
        arith, "fp32-stored" if a.f32_out else "fp64", a.mode.upper(), "pos+vel" if vel_on else "pos only", a.layout)
    if layout == _native.SAT_MAJOR:
        kname = ("k_rows_fast<%s> (branch-free uniform-grid step, one wave per satellite row, lane = time) + k_rows redo pass"
                 if not a.no_fast_path else "k_rows<%s> (one wave per satellite row, lane = time)") % ("pos+vel" if vel_on else "pos")
        if f32_fast:
            kname = kname.replace("k_rows_fast<", "k_rows_fast32<MIXED," if not a.f32_arith else "k_rows_fast32<")
    else:
        kname = ("k_tiles_fast<%s> (16-satellite tiles of lane = time waves, LDS transpose) + k_rows redo pass"
                 if not (a.no_fast_path or a.no_tile_kernel or a.f32_out) else "k_propagate<time-major,%s> (lane = satellite)") % (
                     "pos+vel" if vel_on else "pos")
    out = {
        "metric": ("propagations/sec, 13,478 sats x 1,440 times, at 1/2/4/8 MI355X" if not a.config5_share else
                   "propagations/sec, config 5 (1M sats x 10,000 times, fp32, 8 MI355X): one GPU's 125,000-satellite share"),
        "value": value, "unit": "propagations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
        "scaling": "n/a" if world == 1 else a.scaling, "vs_baseline": None,
        "dtype": "f64" if "fp64 arithmetic" == arith else "f32 (+f64 phase/radius chains)", "data": "synthetic",
        "config": {
            "workload": wl + ("" if a.grid == "uniform" else " [time grid: %s]" % a.grid), "launch_path": dev.last_path(), "n_sats_total": n_total, "n_sats_per_gpu": n_local, "n_times": n_times, "gather": bool(gather),
            "precondition_ms": a.precondition_ms, "precondition_steps": n_pre,
            **({"t_kernel_ms": kernel_only_ms, "t_allgather_ms": max(elapsed / a.steps * 1e3 - kernel_only_ms, 0.0),
                "t_total_ms": elapsed / a.steps * 1e3, "rccl_ranks": world, "chunks": plan.n_chunks,
                "kernel_only_value": props_per_step / (kernel_only_ms / 1e3),
                "t_replicate_ms": replicate_ms, "replicate_value": props_per_step / (replicate_ms / 1e3),
                "replicate_note": "every GPU propagates the FULL catalog itself (no shards, zero bytes moved): the same "
                                  "deliverable as the gathered run -- the full arrays on every GPU",
                "group_host": group_host,
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
            out["secondary"] = run_secondary(torch, _native, synth, cuda, stream, dev, pairs,
                                             skip=tuple(k for k in a.secondary_skip.split(",") if k))
        except Exception as exc:   # never takes the headline down
            out["secondary"] = [{"failed": repr(exc)}]
    # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio, which (redirected to a file or
    # a pipe) sits in the C buffer until the process ends -- flush it first
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if group_host_stuck:
        os._exit(0)   # (a stuck extra must not keep the process alive)


if __name__ == "__main__":
    main()
