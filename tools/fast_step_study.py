#!/usr/bin/env python3
"""CPU study of the branch-free fast step as k_rows_fast runs it (tests/host_emul: the device headers compiled for the
host): acceptance rate of the per-window validation and worst difference from the oracle on a whole catalog.
usage: tools/fast_step_study.py [n_sats] [n_times] [tile] [t0]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from astroz_amd import synth
from oracle import oracle

n_sats = int(sys.argv[1]) if len(sys.argv) > 1 else 13478
n_times = int(sys.argv[2]) if len(sys.argv) > 2 else 1440
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 768
t0 = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 20260926
src = os.path.join(ROOT, "tests", "host_emul", "emul.cpp")
lib = "/tmp/libemul_study.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", lib, src])
E = C.CDLL(lib)
E.emul_init.restype = C.c_uint
E.emul_init.argtypes = [C.c_void_p] * 3
E.emul_rows_fast.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
g = np.array([6378.135, 0.001082616, -0.00000165597, 0.0743669161331734132, -0.00234506972242078, 0.0743669161331734132 * 6378.135 / 60.0])
pairs = synth.synth_catalog(n_near=n_sats, n_deep=0, seed=seed)
tles = [oracle.parse_lines(a, b) for a, b in pairs]
cat = oracle.Catalog(tles, 1)
off = (synth.START_JD - cat.epoch_jd) * 1440.0
times = t0 + np.arange(n_times, dtype=np.float64)
_, p0, v0 = cat.propagate(times, off, layout=oracle.SAT_MAJOR, threads=os.cpu_count())
nf = 85
out = np.zeros((n_times, 6)); bad = np.zeros(n_times, dtype=np.int32)
wr = wv = 0.0; acc = tot = 0; rej_sats = 0
per_class = {0: [0, 0], 1: [0, 0]}
for i, t in enumerate(tles):
    raw = np.array([t.epoch_jd, t.mm_revday, t.ecc, t.incl_deg, t.raan_deg, t.argp_deg, t.ma_deg, t.bstar])
    fields = np.zeros(nf)
    flags = E.emul_init(raw.ctypes.data, g.ctypes.data, fields.ctypes.data)
    ecc = 1 if ((flags >> 12) & 3) else 0
    E.emul_rows_fast(fields.ctypes.data, flags, g.ctypes.data, times[0] + off[i], 1.0, n_times, tile if not ecc else 256, ecc, out.ctypes.data, bad.ctypes.data)
    ok = bad == 0
    tot += n_times; acc += int(ok.sum()); rej_sats += int(not ok.all())
    per_class[ecc][0] += int(ok.sum()); per_class[ecc][1] += n_times
    if ok.any():
        wr = max(wr, np.abs(out[ok, :3] - p0[i][ok]).max()); wv = max(wv, np.abs(out[ok, 3:] - v0[i][ok]).max())
print("accepted %.4f %% of %d points; satellites with a rejected segment: %d of %d" % (100.0 * acc / tot, tot, rej_sats, len(tles)))
for c in (0, 1):
    if per_class[c][1]:
        print("  %s form: %.4f %% of %d" % ("eccentric" if c else "near-circular", 100.0 * per_class[c][0] / per_class[c][1], per_class[c][1]))
print("max |dr| = %.3e km, max |dv| = %.3e km/s over the accepted points" % (wr, wv))
