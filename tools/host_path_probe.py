"""PCIe-inclusive rates of the host-pointer API (what numpy callers of the reference-compatible
Python layer see).  Run on the GPU box; prints one JSON object."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import astroz_amd
from astroz_amd import synth, _native
from astroz_amd.api import SatrecArray, Satrec, WGS72
from datetime import datetime, timezone

out = {}
pairs = synth.synth_catalog(13478, 0)
text = synth.pairs_to_text(pairs)
t0 = time.perf_counter(); c = astroz_amd.Constellation(text); out["Constellation_from_text_s"] = time.perf_counter() - t0
times = np.arange(1440.0)
start = datetime(2025, 5, 5, tzinfo=timezone.utc)
for name, kw in (("propagate_ecef_pos", dict()), ("propagate_teme_posvel", dict(output="teme", velocities=True))):
    astroz_amd.propagate(c, times[:64], start_time=start, **kw)
    t0 = time.perf_counter(); r = astroz_amd.propagate(c, times, start_time=start, **kw); dt = time.perf_counter() - t0
    nbytes = sum(a.nbytes for a in (r if isinstance(r, tuple) else (r,)))
    out[name] = {"s": dt, "mprops_per_s": 13478 * 1440 / dt / 1e6, "GB_per_s": nbytes / dt / 1e9}
# raw C-ABI host call into preallocated arrays (second call: pages already touched)
dev = c._dev
pos = np.empty((1440, dev.n, 3)); vel = np.empty_like(pos)
off = (synth.START_JD - dev.epochs) * 1440.0
for rep in range(2):
    t0 = time.perf_counter(); dev.propagate_host(times, off, pos=pos, vel=vel); dt = time.perf_counter() - t0
out["azh_propagate_host_posvel_touched"] = {"s": dt, "mprops_per_s": 13478 * 1440 / dt / 1e6, "GB_per_s": (pos.nbytes + vel.nbytes) / dt / 1e9}
print(json.dumps(out, indent=1))
