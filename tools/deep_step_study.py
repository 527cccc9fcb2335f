#!/usr/bin/env python3
"""CPU study of the deep-space device step (tests/host_emul) against the oracle on config 3's deep members."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from astroz_amd import synth
from oracle import oracle
n_deep = int(sys.argv[1]) if len(sys.argv) > 1 else 1522
n_times = int(sys.argv[2]) if len(sys.argv) > 2 else 1440
src = os.path.join(ROOT, "tests", "host_emul", "emul.cpp"); lib = "/tmp/libemul_study.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", lib, src])
E = C.CDLL(lib); E.emul_init.restype = C.c_uint; E.emul_init.argtypes = [C.c_void_p] * 3
E.emul_propagate.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
g = np.array([6378.135, 0.001082616, -0.00000165597, 0.0743669161331734132, -0.00234506972242078, 0.0743669161331734132 * 6378.135 / 60.0])
pairs = synth.synth_catalog(n_near=0, n_deep=n_deep, seed=20260926)
tles = [oracle.parse_lines(a, b) for a, b in pairs]
cat = oracle.Catalog(tles, 1)
off = (synth.START_JD - cat.epoch_jd) * 1440.0
times = np.arange(n_times, dtype=np.float64)
e0, p0, v0 = cat.propagate(times, off, layout=oracle.SAT_MAJOR, threads=os.cpu_count())
wr = wv = 0.0; worst = None; nlyd = 0
for i, t in enumerate(tles):
    raw = np.array([t.epoch_jd, t.mm_revday, t.ecc, t.incl_deg, t.raan_deg, t.argp_deg, t.ma_deg, t.bstar])
    fields = np.zeros(85); flags = E.emul_init(raw.ctypes.data, g.ctypes.data, fields.ctypes.data)
    tt = times + off[i]; out = np.zeros((n_times, 6)); rc = np.zeros(n_times, dtype=np.int32)
    E.emul_propagate(fields.ctypes.data, flags, g.ctypes.data, tt.ctypes.data, n_times, 1, out.ctypes.data, rc.ctypes.data)
    assert np.array_equal(rc.astype(np.uint8), e0[i]), i
    dr = np.abs(out[:, :3] - p0[i]).max(); dv = np.abs(out[:, 3:] - v0[i]).max()
    nlyd += t.incl_deg < 11.46
    if dr > wr: wr, worst = dr, (i, t.incl_deg, t.ecc, t.mm_revday)
    wv = max(wv, dv)
print("deep members %d (%d near-equatorial): max |dr| = %.3e km, max |dv| = %.3e km/s; worst %s" % (len(tles), nlyd, wr, wv, worst))
