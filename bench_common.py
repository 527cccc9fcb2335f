"""Shared pieces of bench.py / bench_secondary.py: the algorithmic-traffic constants, the CPU baseline leg, the
power / clock sampler, the kernel-source fingerprint, and `compact_line` -- the ONE short JSON line the driver parses."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

BYTES_OUT_PV = 48.0          # 6 fp64 written
BYTES_OUT_P = 24.0
ELEM_BYTES_PER_SAT = 32 * 8 + 8 + 4   # element rows read per satellite per time tile + offset + flags
FLOPS_PER_PROP = 581.0       # reference formulation at K = 4 Newton trips (405 + 44 K)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz



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


def cpu_baseline_sample(pairs, times, offsets, seconds, sat_major, max_sats=13478, max_times=1440):
    """cpu_baseline on a BOUNDED sample for workloads too large to restate on the host (config 5's 125,000 x 10,000 share is
    60 GB of fp64 results): the first `max_sats` satellites x the first `max_times` grid points of the same catalog and grid."""
    n_s, n_t = min(len(pairs), max_sats), min(len(times), max_times)
    cb, _ = cpu_baseline(pairs[:n_s], np.ascontiguousarray(times[:n_t]), np.ascontiguousarray(offsets[:n_s]), seconds, sat_major)
    if n_s < len(pairs) or n_t < len(times):
        cb["sample"] = "SUBSET of the rank-0 workload (first %d of %d satellites x first %d of %d times): " % (n_s, len(pairs), n_t, len(times)) + cb["sample"]
    return cb


def rank_certificate(torch, pairs, times, offsets, pos, vel, sat_major, rank, mode=0, ref_jd=0.0, n_rows=24, threads=None):
    """What ONE rank can certify about its own output without any other rank ("no gather -- verify by
    checksums + sampled rows"): `n_rows` rows spread over the rank's catalog, every time, against the fp64 oracle; a checksum
    (fp64 sum and largest magnitude) and a finiteness flag over EVERY element the rank wrote.  pos / vel: the rank's device
    arrays, (n_local [+ padding], n_times, 3) satellite-major or (n_times, >= n_local, 3) time-major; fp64 or fp32."""
    from oracle import oracle
    n_local = len(pairs)
    cert = {"rank": int(rank), "n_sats": int(n_local)}
    if n_local == 0:
        return cert
    rows = np.unique(np.linspace(0, n_local - 1, n_rows).astype(np.int64))
    cat = oracle.Catalog.from_pairs([pairs[i] for i in rows], oracle.WGS72)
    thr = threads or usable_cpus()
    _, p0, v0 = cat.propagate(times, np.ascontiguousarray(offsets[rows]), mode=mode, reference_jd=ref_jd, layout=oracle.SAT_MAJOR, threads=thr)
    idx = torch.as_tensor(rows, device=pos.device)

    def take(a):
        return (a[idx] if sat_major else a[:, idx].permute(1, 0, 2)).cpu().numpy().astype(np.float64)
    d = take(pos) - p0
    if mode == 2:
        d[..., 1] = (d[..., 1] + np.pi) % (2 * np.pi) - np.pi
    cert.update({"sample_sats": int(len(rows)), "max_dr": float(np.abs(d).max())})
    own = pos[:n_local] if sat_major else pos[:, :n_local]
    cert["checksum"] = float(torch.sum(own, dtype=torch.float64))
    cert["absmax"] = float(max(own.max(), -own.min()))      # (no |x| temporary: the config-5 share is 15 GB per array)
    finite = bool(torch.isfinite(own).all())
    if vel is not None:
        cert["max_dv"] = float(np.abs(take(vel) - v0).max())
        ownv = vel[:n_local] if sat_major else vel[:, :n_local]
        cert["checksum"] += float(torch.sum(ownv, dtype=torch.float64))
        finite = finite and bool(torch.isfinite(ownv).all())
    cert["finite"] = finite
    return cert


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


def describe_workload(a, world, sharded, gather, n_chunks, n_total, n_times, sat_major, mode, vel_on):
    """(config.workload, config.parallelism, the arithmetic behind the outputs, roofline.kernel) of a bench.py invocation"""
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
            ", RCCL all-gather of the full result onto every GPU (%d-chunk compute/gather pipeline)" % n_chunks
            if gather else ", NO gather (--no-gather: every GPU keeps its shard)")
        par = "satellite-sharded x%d%s" % (world, " + RCCL all-gather" if gather else ", no data-path collective")
    else:
        wl = "WEAK scaling (--scaling weak): %d GPUs x an own %d-sat synthetic catalog x %d one-minute steps, no gather" % (
            world, a.sats + a.deep, n_times)
        par = "independent catalogs x%d, no data-path collective" % world
    f32_fast = a.f32_out and not a.f32_fp64 and not a.no_fast_path and sat_major and mode == 0
    arith = "fp64 arithmetic" if not f32_fast else (
        "packed fp32 arithmetic with fp64 phase and radius chains (near-circular members; fp64 for the rest)" if a.f32_arith else
        "mixed-precision arithmetic (O(1) quantities fp64, small ones packed fp32; near-circular members; fp64 for the rest)")
    wl += ", %s, %s %s %s, %s-major device-resident output" % (
        arith, "fp32-stored" if a.f32_out else "fp64", a.mode.upper(), "pos+vel" if vel_on else "pos only", a.layout)
    if sat_major:
        kname = ("k_rows_fast<%s> (branch-free uniform-grid step, one wave per satellite row, lane = time) + k_rows redo pass"
                 if not a.no_fast_path else "k_rows<%s> (one wave per satellite row, lane = time)") % ("pos+vel" if vel_on else "pos")
        if f32_fast:
            kname = kname.replace("k_rows_fast<", "k_rows_fast32<MIXED," if not a.f32_arith else "k_rows_fast32<")
    else:
        kname = ("k_tiles_fast<%s> (16-satellite tiles of lane = time waves, LDS transpose) + k_rows redo pass"
                 if not (a.no_fast_path or a.no_tile_kernel or a.f32_out) else "k_propagate<time-major,%s> (lane = satellite)") % (
                     "pos+vel" if vel_on else "pos")
    return wl, par, arith, kname


LINE_LIMIT = 4096   # bytes: the driver keeps a bounded tail of stdout and parses the LAST line (round 4's 21-KB line was cut)


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _num(x, sig=6):
    """floats to `sig` significant digits (the full-precision values are in the full record)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    return x


def compact_line(out, full_path=None, limit=LINE_LIMIT):
    """The final stdout line of bench.py: ONE compact JSON object below `limit` bytes with exactly the fields the contract
    names -- metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data,
    config{workload, ...}, roofline{...}, cpu_baseline{...}, parity -- plus `secondary_summary` (key -> [ms_per_step, frac])
    and the path of the full record.  Free text is cut to a fixed length; if the line is still too long the optional
    parts are dropped in a fixed order (never the contract fields)."""
    cfg = out.get("config") or {}
    rf = out.get("roofline") or {}
    cb = out.get("cpu_baseline")
    line = {k: _num(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                          "scaling", "vs_baseline", "dtype", "data")}
    c = {"workload": _short(cfg.get("workload", ""), 420)}
    for k in ("n_sats_total", "n_sats_per_gpu", "n_times", "parallelism", "gather", "launch_path", "t_kernel_ms", "t_allgather_ms",
              "t_total_ms", "kernel_only_value", "t_kernel_graphs_ms", "t_replicate_ms", "replicate_value", "chunks", "rccl_ranks", "gather_bytes_per_gpu"):
        if k in cfg:
            c[k] = _num(cfg[k]) if not isinstance(cfg[k], str) else _short(cfg[k], 120)
    ss = cfg.get("sharded_screen")
    if isinstance(ss, dict):
        c["sharded_screen"] = {k: _num(ss[k]) for k in ("ms", "value", "index_mismatches", "max_dd_km") if k in ss} or \
            {"failed": _short(ss.get("failed", "?"), 80)}
    gh = cfg.get("group_host")
    if isinstance(gh, dict):
        c["group_host"] = {k: _num(gh[k]) for k in ("ms_per_call", "devices", "value", "GB_per_s") if k in gh} or \
            {"failed": _short(gh.get("failed", "?"), 80)}
    line["config"] = c
    line["roofline"] = {k: (_short(rf[k], 200) if isinstance(rf.get(k), str) else _num(rf.get(k)))
                        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
                                  "algorithmic_bytes_per_launch")}
    if isinstance(cb, dict):
        t1 = cb.get("threads_1")
        line["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "threads_1": _num(t1.get("value")) if isinstance(t1, dict) else None,
                                "cpu_model": _short(cb.get("cpu_model", ""), 60), "sample": _short(cb.get("sample", ""), 220)}
    if "parity" in out:
        par = out["parity"]
        if isinstance(par, dict):
            line["parity"] = {k: (_num(v) if not isinstance(v, str) else _short(v, 100)) for k, v in par.items() if k != "per_rank"}
            if isinstance(par.get("per_rank"), list):   # one entry per rank: [rank, max_dr, max_dv, checksum, finite]
                line["parity"]["per_rank"] = [[c.get("rank"), _num(c.get("max_dr"), 3), _num(c.get("max_dv"), 3), _num(c.get("checksum"), 12),
                                               c.get("finite")] for c in par["per_rank"] if isinstance(c, dict)]
        else:
            line["parity"] = par
    vi = out.get("valu_issue")
    if isinstance(vi, dict):
        line["valu_issue"] = {k: _num(vi[k], 4) for k in ("issue_slot_frac", "valu_insts_per_propagation", "sclk_mhz") if k in vi}
    pw = out.get("power")
    if isinstance(pw, dict):
        line["power"] = {k: _num(pw[k], 4) for k in ("socket_w_median", "sclk_mhz_median", "socket_w_limit") if k in pw}
    sec = out.get("secondary")
    if isinstance(sec, list):
        summ = {}
        for e in sec:
            if not isinstance(e, dict) or "key" not in e:
                continue
            if "failed" in e:
                summ[e["key"]] = "failed"
                continue
            ms = e.get("ms_per_step", e.get("ms_per_call"))
            fr = (e.get("roofline") or {}).get("frac")
            summ[e["key"]] = [_num(ms, 4) if ms is not None else None, _num(fr, 3) if fr is not None else None]
        line["secondary_summary"] = summ
    if full_path:
        line["full_record"] = full_path
    for drop in (None, "power", "valu_issue", "secondary_summary"):
        if drop:
            line.pop(drop, None)
        txt = json.dumps(line, separators=(",", ":"))
        if len(txt.encode()) < limit:
            return txt
    line["config"] = {"workload": _short(c["workload"], 200)}
    return json.dumps(line, separators=(",", ":"))
