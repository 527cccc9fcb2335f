"""Where one satellite x 10^7 times through the constellation call loses time: variants of rows x times x span."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import torch
from astroz_amd import synth, _native
out = {}
pairs = synth.synth_catalog(13478, 0)
cuda = torch.device("cuda", 0)
def run(key, rows, n, span, exact=False, tile=0):
    dev = _native.DeviceConstellation.from_tle_lines([pairs[i] for i in rows], 1, 0)
    if tile:
        dev.set_time_tile(tile, tile)
    times = np.arange(n) * (span / n) if exact else np.linspace(0.0, span, n)
    off = np.zeros(len(rows))
    pos = torch.empty((len(rows), n, 3), dtype=torch.float64, device=cuda)
    vel = torch.empty_like(pos)
    dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=_native.SAT_MAJOR)
    dev.synchronize()
    ms = []
    for _ in range(10):
        dev.propagate_device_cached(pos.data_ptr(), vel.data_ptr(), layout=_native.SAT_MAJOR)
        dev.synchronize()
        ms.append(dev.last_kernel_ms())
    out[key] = {"ms": float(np.median(ms)), "path": dev.last_path(), "G_props_s": len(rows) * n / np.median(ms) / 1e6}
run("8x1.25e6", list(range(8)), 1_250_000, 1440.0, exact=True)
run("128x78125", list(range(128)), 78_125, 1440.0, exact=True)
run("1024x9766", list(range(1024)), 9_766, 1440.0, exact=True)
run("4096x2441", list(range(4096)), 2_441, 1440.0, exact=True)
run("13478x742", list(range(13478)), 742, 742.0, exact=True)
run("13478x1440", list(range(13478)), 1440, 1440.0, exact=True)
run("8x1.25e6_tile256", list(range(8)), 1_250_000, 1440.0, exact=True, tile=256)
run("8x1.25e6_tile1536", list(range(8)), 1_250_000, 1440.0, exact=True, tile=1536)
print(json.dumps(out, indent=1))
