import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g; g.build()
from astroz_amd import _native as native, synth
from oracle import oracle as orc
pairs = synth.synth_catalog(n_near=200, n_deep=40, seed=13)
dev = native.DeviceConstellation.from_tle_lines(pairs, 0, 0)
cat = orc.Catalog.from_pairs(pairs, 0)
times = np.arange(0.0, 300.0, 5.0)
ref = synth.START_JD + 0.25
off = (ref - dev.epochs) * 1440.0
err, deep, irez = dev.status
ecc = np.array([cat.field(i, "ecco") for i in range(cat.n)])
for fast in (True, False):
    dev.set_fast_path(fast)
    for rep in range(2):
        pos = np.empty((dev.n, len(times), 3)); vel = np.empty_like(pos)
        dev.propagate_host(times, off, pos=pos, vel=vel, mode=native.OUT_GEODETIC, reference_jd=ref, layout=native.SAT_MAJOR)
        _, p0, v0 = cat.propagate(times, off, mode=orc.GEODETIC, reference_jd=ref, layout=orc.SAT_MAJOR)
        dlon = pos[..., 1] - p0[..., 1]
        w = np.abs((dlon + np.pi) % (2*np.pi) - np.pi).max(axis=1)
        badrows = np.flatnonzero(w > 1e-10)
        print("fast", fast, "rep", rep, "bad rows:", [(int(r), bool(deep[r]), float(ecc[r]), float(w[r])) for r in badrows[:12]], len(badrows))
        print("   lat max", np.abs(pos[...,0]-p0[...,0]).max(), "alt max", np.abs(pos[...,2]-p0[...,2]).max(), "vel", np.abs(vel-v0).max())
        if len(badrows):
            r = badrows[0]
            print("   row", r, "dev lon", pos[r,:5,1], "orc lon", p0[r,:5,1], "dev lat", pos[r,:3,0], "orc lat", p0[r,:3,0])
