#!/usr/bin/env python3
"""Measured HIP-vs-oracle differences on the long-span grids of tests/test_gpu_parity.py, split by population, next to the
oracle's own sensitivity to one ulp of the time argument (what two correct fp64 evaluations may differ by)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
from astroz_amd import _native as native, synth
from oracle import oracle as orc

out = {}
def run(name, pairs, times, off):
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)
    for lay, olay, tag in ((native.SAT_MAJOR, orc.SAT_MAJOR, "sat"), (native.TIME_MAJOR, orc.TIME_MAJOR, "time")):
        shape = (dev.n, len(times), 3) if lay == native.SAT_MAJOR else (len(times), dev.n, 3)
        pos = np.empty(shape); vel = np.empty(shape); err = np.zeros((dev.n, len(times)), dtype=np.uint8)
        dev.propagate_host(times, off, pos=pos, vel=vel, layout=lay, err=err)
        e0, p0, v0 = cat.propagate(times, off, layout=olay, threads=8)
        # the oracle one ulp of time later
        tu = np.nextafter(times, np.inf)
        _, p1, v1 = cat.propagate(tu, off, layout=olay, threads=8)
        if lay == native.TIME_MAJOR:
            pos, vel, p0, v0, p1, v1 = (np.transpose(a, (1, 0, 2)) for a in (pos, vel, p0, v0, p1, v1))
        ok = (e0 == 0)[:, :, None]
        deep = cat.is_deep
        for pop, m in (("near", ~deep), ("deep", deep)):
            if not m.any():
                continue
            dtv = 7.5 * 0  # placeholder
            out["%s/%s/%s" % (name, tag, pop)] = {
                "max_dr_km": float(np.abs((pos - p0) * ok)[m].max()), "max_dv_kms": float(np.abs((vel - v0) * ok)[m].max()),
                # the motion during one ulp of time is removed: what remains is the oracle's own rounding noise
                "oracle_ulp_dr_km": float(np.abs(((p1 - p0) - v0 * ((tu - times) * 60.0)[None, :, None]) * ok)[m].max()),
            }

pairs = synth.synth_catalog(n_near=90, n_deep=10, seed=77)
dev0 = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
run("uniform_10000", pairs, np.arange(10000, dtype=np.float64), (synth.START_JD - dev0.epochs) * 1440.0)
pairs = synth.synth_catalog(n_near=300, n_deep=120, seed=8)
rng = np.random.default_rng(5)
times = np.concatenate([np.linspace(-20000, 20000, 97), rng.uniform(-20000, 20000, 60), np.arange(-700.0, 900.0, 10.0)])
run("pm2weeks", pairs, times, None)
print(json.dumps(out, indent=1))
