"""One satellite x 10^7 uniformly spaced times through the CONSTELLATION call of a 1-satellite handle (row kernels, window plan)
against azh_propagate_one_device (k_one_satellite, any times).  Run on the GPU box; prints one JSON object."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import torch
from astroz_amd import synth, _native
from oracle import oracle

out = {}
pairs = synth.synth_catalog(13478, 0)
cuda = torch.device("cuda", 0)
for name, idx in (("leo", 0), ("ecc", None)):
    if idx is None:
        dev_all = _native.DeviceConstellation.from_tle_lines(pairs[:2000], 1, 0)
        ecc = dev_all.field("ecco")
        idx = int(np.argmax(ecc))
    dev = _native.DeviceConstellation.from_tle_lines([pairs[idx]], 1, 0)
    for n, span in ((10_000_000, 14400.0), (1_000_000, 1440.0 * 7)):
        times = np.linspace(0.0, span, n)
        off = np.zeros(1)
        pos = torch.empty((1, n, 3), dtype=torch.float64, device=cuda)
        vel = torch.empty_like(pos)
        t0 = time.perf_counter()
        dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=_native.SAT_MAJOR)
        dev.synchronize()
        first = (time.perf_counter() - t0) * 1e3
        ms = []
        for _ in range(10):
            dev.propagate_device_cached(pos.data_ptr(), vel.data_ptr(), layout=_native.SAT_MAJOR)
            dev.synchronize()
            ms.append(dev.last_kernel_ms())
        ts = torch.as_tensor(times, device=cuda)
        po = torch.empty((n, 3), dtype=torch.float64, device=cuda)
        ve = torch.empty_like(po)
        ms1 = []
        for _ in range(6):
            dev.propagate_one_device(0, ts.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, None)
            dev.synchronize()
            ms1.append(dev.last_kernel_ms())
        pick = np.unique(np.linspace(0, n - 1, 4096).astype(np.int64))
        cat = oracle.Catalog.from_pairs([pairs[idx]], oracle.WGS72)
        _, p0, v0 = cat.propagate(times[pick], None, layout=oracle.SAT_MAJOR)
        sel = torch.as_tensor(pick, device=cuda)
        out["%s_%d" % (name, n)] = {
            "grid_call_kernel_ms": float(np.median(ms)), "first_call_wall_ms": first, "path": dev.last_path(),
            "one_device_kernel_ms": float(np.median(ms1[1:])),
            "grid_vs_oracle_km": float(np.abs(pos[0][sel].cpu().numpy() - p0[0]).max()),
            "grid_vs_oracle_kms": float(np.abs(vel[0][sel].cpu().numpy() - v0[0]).max()),
            "one_vs_oracle_km": float(np.abs(po[sel].cpu().numpy() - p0[0]).max())}
        del pos, vel, ts, po, ve
print(json.dumps(out, indent=1))
