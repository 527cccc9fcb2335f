import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from astroz_amd import _native as native, synth
pairs = synth.synth_catalog(n_near=1203, n_deep=131, seed=52)
for k in (7, 640):
    l1, l2 = pairs[k]; pairs[k] = (l1, l2[:52] + "17.80000000" + l2[63:])
dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
err, deep, irez = dev.status
rng = np.random.default_rng(16)
n_times = 333
times = np.sort(np.arange(n_times, dtype=np.float64) * 2.0 + rng.uniform(-0.9, 0.9, n_times)) - 77.0
off = (synth.START_JD - dev.epochs) * 1440.0
for stride in (dev.n, dev.n + 5):
    pos = torch.full((n_times, stride, 3), float("nan"), dtype=torch.float64, device="cuda")
    v = torch.full_like(pos, float("nan"))
    torch.cuda.synchronize()
    dev.propagate_device(times, off, pos.data_ptr(), v.data_ptr(), layout=native.TIME_MAJOR, stride=stride)
    dev.synchronize()
    got = pos.cpu().numpy()[:, :dev.n]
    bad = np.isnan(got).any(axis=2)
    print("stride", stride, "path", dev.last_path(), "nan count", bad.sum(), "of", bad.size)
    if bad.any():
        ts, ss = np.nonzero(bad)
        print(" sats with NaN:", np.unique(ss)[:40], "n", len(np.unique(ss)))
        print(" deep?", deep[np.unique(ss)][:40], "err", err[np.unique(ss)][:40])
        print(" times with NaN:", np.unique(ts)[:40], len(np.unique(ts)))
from oracle import oracle as orc
cat = orc.Catalog.from_pairs(pairs, 1)
stride = dev.n + 5
err_t = torch.zeros((dev.n, n_times), dtype=torch.uint8, device="cuda")
pos = torch.full((n_times, stride, 3), float("nan"), dtype=torch.float64, device="cuda")
v = torch.full_like(pos, float("nan"))
torch.cuda.synchronize()
dev.set_tile_kernel(16)
dev.propagate_device(times, off, pos.data_ptr(), v.data_ptr(), mode=0, reference_jd=synth.START_JD, mask=None, layout=native.TIME_MAJOR, stride=stride, d_err=err_t.data_ptr())
dev.synchronize()
e0, p0, v0 = cat.propagate(times, off, layout=orc.TIME_MAJOR, velocities=True, mode=0, reference_jd=synth.START_JD)
got = pos.cpu().numpy()
print("got nan", np.isnan(got[:, :dev.n]).sum(), "p0 nan", np.isnan(p0).sum(), "p0 shape", p0.shape)
if np.isnan(p0).any():
    ts, ss, _ = np.nonzero(np.isnan(p0)); print(np.unique(ss), deep[np.unique(ss)], err[np.unique(ss)], e0[np.unique(ss)][:, :3])
