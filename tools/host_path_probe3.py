"""Host-returning call, fresh arrays: what the page-touch threads buy, and what keeping the previous result alive costs.
Run on the GPU box; prints one JSON object."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from astroz_amd import synth, _native

out = {"cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
try:
    out["cpu.max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    out["memory.max"] = open("/sys/fs/cgroup/memory.max").read().strip()
except OSError:
    pass
pairs = synth.synth_catalog(13478, 0)
dev = _native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
n, T = dev.n, 1440
times = np.arange(float(T))
off = (synth.START_JD - dev.epochs) * 1440.0


def fresh_call():
    pos = np.empty((T, n, 3)); vel = np.empty_like(pos)
    t0 = time.perf_counter(); dev.propagate_host(times, off, pos=pos, vel=vel); dt = time.perf_counter() - t0
    return dt * 1e3, pos, vel


for thr in (0, 2, 4, 6, 8, 12, 16):
    _native.set_host_copy_threads(thr)
    a = []
    for _ in range(4):           # previous result dropped before the next call
        dt, p, v = fresh_call(); a.append(dt); del p, v
    b = []
    keep = None
    for _ in range(4):           # previous result still alive during the next call
        dt, p, v = fresh_call(); b.append(dt); keep = (p, v)
    del keep
    out["threads_%d" % thr] = {"drop_prev_ms": a, "keep_prev_ms": b}
_native.set_host_copy_threads(-1)
pos = np.empty((T, n, 3)); vel = np.empty_like(pos)
dev.propagate_host(times, off, pos=pos, vel=vel)
c = []
for _ in range(4):
    t0 = time.perf_counter(); dev.propagate_host(times, off, pos=pos, vel=vel); c.append((time.perf_counter() - t0) * 1e3)
out["touched_ms"] = c
print(json.dumps(out, indent=1))
