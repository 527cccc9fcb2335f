#!/usr/bin/env python3
"""Per-population cost of the deep-space kernel (development probe; run on the GPU box)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astroz_amd import _native, synth

LAYOUT = _native.TIME_MAJOR if "time" in sys.argv[1:] else _native.SAT_MAJOR

def run(name, el, n_times=1440, reps=10):
    pairs = synth.elements_to_pairs(el, 1)
    dev = _native.DeviceConstellation.from_tle_lines(pairs, _native.WGS72, 0)
    e, d, r = dev.status
    times = np.arange(n_times, dtype=np.float64)
    off = (synth.START_JD - dev.epochs) * 1440.0
    pos = torch.empty((n_times, dev.n, 3), dtype=torch.float64, device="cuda")
    vel = torch.empty_like(pos)
    dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=LAYOUT)
    dev.synchronize()
    ms = []
    for _ in range(5):
        dev.propagate_device_cached(pos.data_ptr(), vel.data_ptr(), layout=LAYOUT)
    for _ in range(reps):
        dev.propagate_device_cached(pos.data_ptr(), vel.data_ptr(), layout=LAYOUT)
        dev.synchronize()
        ms.append(dev.last_kernel_ms())
    print("%-10s n=%5d deep=%5d irez=%s  kernel %.3f ms  (%.2f G props/s)" % (
        name, dev.n, int(d.sum()), np.bincount(r, minlength=3).tolist(), np.median(ms), dev.n * n_times / np.median(ms) / 1e6))

rng = np.random.default_rng(1)
n = 1536
base = synth.deep_space_elements(20000, seed=3)
def pick(mask):
    idx = np.flatnonzero(mask)[:n]
    return {k: v[idx] for k, v in base.items()}
mm, ecc, inc = base["mm"], base["ecc"], base["incl"]
run("GEO", pick((mm < 1.1)))
run("GEO_hiinc", pick((mm < 1.1) & (inc > 12.5)))
run("GNSS", pick((mm > 1.6) & (mm < 2.2) & (ecc < 0.03)))
run("Molniya", pick((ecc > 0.59) & (mm > 1.99) & (mm < 2.02)))
run("GTO", pick((ecc >= 0.3) & ~((mm > 1.99) & (mm < 2.02) & (ecc > 0.59))))
run("other", pick((mm >= 3.0) & (ecc < 0.11)))
run("mix", {k: v[:n] for k, v in base.items()})
run("mix6064", {k: v[:6064] for k, v in base.items()})
ne = synth.near_earth_elements(n, seed=4)
run("near", ne)
