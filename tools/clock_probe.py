#!/usr/bin/env python3
"""Per-launch duration over seconds of sustained load, with the device's clock / power state sampled from
sysfs beside it (development tool; run on the GPU box).
  python tools/clock_probe.py [--seconds 2.0] [--no-fast-path] [--pos-only] [--layout time]"""
import argparse, glob, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sysfs_sampler(stop, out, period=0.005):
    cards = sorted(glob.glob("/sys/class/drm/card*/device"))
    files = {}
    for c in cards:
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent"):
            p = os.path.join(c, name)
            if os.path.exists(p):
                files[name] = p
        for h in glob.glob(os.path.join(c, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(c, "hwmon/hwmon*/power1_input")) + \
                glob.glob(os.path.join(c, "hwmon/hwmon*/temp*_input")) + glob.glob(os.path.join(c, "hwmon/hwmon*/freq*_input")):
            files[os.path.basename(h)] = h
        if files:
            break
    out["files"] = files
    t0 = time.perf_counter()
    while not stop.is_set():
        rec = {"t": time.perf_counter() - t0}
        for k, p in files.items():
            try:
                txt = open(p).read()
                if k.startswith("pp_dpm"):
                    cur = [ln for ln in txt.splitlines() if "*" in ln]
                    rec[k] = cur[0].split(":")[1].strip().rstrip("*").strip() if cur else txt.strip()[:40]
                else:
                    rec[k] = txt.strip()
            except OSError:
                pass
        out.setdefault("samples", []).append(rec)
        time.sleep(period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--no-fast-path", action="store_true")
    ap.add_argument("--pos-only", action="store_true")
    ap.add_argument("--layout", default="sat")
    ap.add_argument("--idle-ms", type=float, default=0.0, help="host sleep between launches (duty-cycle experiments)")
    ap.add_argument("--tag", default="probe")
    a = ap.parse_args()
    import torch
    from astroz_amd import _native, synth
    pairs = synth.synth_catalog(n_near=13478, n_deep=0, seed=20260926)
    dev = _native.DeviceConstellation.from_tle_lines(pairs, _native.WGS72, 0)
    if a.no_fast_path:
        dev.set_fast_path(False)
    times = np.arange(1440, dtype=np.float64)
    off = (synth.START_JD - dev.epochs) * 1440.0
    layout = _native.TIME_MAJOR if a.layout == "time" else _native.SAT_MAJOR
    shape = (1440, dev.n, 3) if layout == _native.TIME_MAJOR else (dev.n, 1440, 3)
    pos = torch.empty(shape, dtype=torch.float64, device="cuda")
    vel = None if a.pos_only else torch.empty(shape, dtype=torch.float64, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    pp, vp = pos.data_ptr(), (None if vel is None else vel.data_ptr())
    dev.propagate_device(times, off, pp, vp, layout=layout, stream=sp)
    torch.cuda.synchronize()
    dev.set_timing(False)
    time.sleep(1.0)  # start from an idle device
    stop, smp = threading.Event(), {}
    th = threading.Thread(target=sysfs_sampler, args=(stop, smp), daemon=True)
    th.start()
    time.sleep(0.05)
    n = int(a.seconds / 0.00025)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t_host0 = time.perf_counter()
    evs[0].record(stream)
    for i in range(n):
        dev.propagate_device_cached(pp, vp, layout=layout, stream=sp)
        evs[i + 1].record(stream)
        if a.idle_ms:
            torch.cuda.synchronize(); time.sleep(a.idle_ms / 1e3)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_host0
    stop.set(); th.join()
    d = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n)])
    tt = np.cumsum(d)
    print("tag=%s launches=%d wall=%.3fs  mean=%.4f ms  median=%.4f  p10=%.4f p90=%.4f" % (a.tag, n, wall, d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    # duration vs elapsed time, in 25-ms bins
    edges = np.arange(0, tt[-1] + 25, 25.0)
    idx = np.digitize(tt, edges)
    line = []
    for b in range(1, len(edges)):
        sel = d[idx == b]
        if len(sel):
            line.append("%.0f:%.3f" % (edges[b - 1], np.median(sel)))
    print("median launch ms per 25-ms bin:", " ".join(line))
    ss = smp.get("samples", [])
    print("sysfs files:", list(smp.get("files", {}).keys()))
    for rec in ss[:: max(1, len(ss) // 40)]:
        print("  ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in rec.items()})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"durations_ms": d.tolist(), "sysfs": ss}, open(os.path.join(ROOT, "gpurun_out", "clock_%s.json" % a.tag), "w"))


if __name__ == "__main__":
    main()
