#!/usr/bin/env python3
"""CPU study of the fp32-output steps (host emulation, tests/host_emul): error of the packed fp32 step, the mixed-precision
step and fp64-rounded-at-the-store against the fp64 oracle on the synthetic catalog (near-circular members, 10,000-minute
span).  tools/f32_step_study.py [n_sats]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from astroz_amd import synth
from oracle import oracle as orc

src = os.path.join(ROOT, "tests", "host_emul", "emul.cpp")
lib = "/tmp/libemul_study.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", lib, src])
E = C.CDLL(lib)
E.emul_init.restype = C.c_uint
E.emul_init.argtypes = [C.c_void_p] * 3
for f in (E.emul_propagate_fast32, E.emul_propagate_fast32p):
    f.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
E.emul_num_fields.restype = C.c_int
orc.build()
n_sats = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pairs = synth.synth_catalog(n_sats, 0, seed=19)
tles = [orc.parse_lines(a, b) for a, b in pairs]
cat = orc.Catalog(tles, 1)
gr = orc.grav_constants(1) if hasattr(orc, "grav_constants") else None
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_host_cpu as T
g = T._grav6(1)
nf = E.emul_num_fields()
n, step, lane_steps = 79, 1.0, 128
off = (synth.START_JD - cat.epoch_jd) * 1440.0
res = {k: [0.0, 0.0, 0] for k in ("packed", "mixed", "rounded")}
for i, t in enumerate(tles):
    raw = np.array([t.epoch_jd, t.mm_revday, t.ecc, t.incl_deg, t.raan_deg, t.argp_deg, t.ma_deg, t.bstar])
    fields = np.zeros(nf)
    flags = E.emul_init(raw.ctypes.data, g.ctypes.data, fields.ctypes.data)
    outs = {}
    for key, fn in (("packed", E.emul_propagate_fast32), ("mixed", E.emul_propagate_fast32p)):
        out = np.zeros((n, 2, 6)); bad = np.zeros(n, dtype=np.int32)
        fn(fields.ctypes.data, flags, g.ctypes.data, off[i], step, lane_steps, n, out.ctypes.data, bad.ctypes.data)
        outs[key] = (out, bad)
    for k in range(0, n, 2):
        for half in (0, 1):
            _, r, v = cat.propagate_one(i, off[i] + k * step * lane_steps + half * step)
            rr = np.concatenate([r, v])
            rounded = rr.astype(np.float32).astype(np.float64)
            for key in ("packed", "mixed"):
                out, bad = outs[key]
                if bad[k]:
                    continue
                e = res[key]
                e[0] = max(e[0], np.linalg.norm(out[k, half, :3] - r)); e[1] = max(e[1], np.linalg.norm(out[k, half, 3:] - v)); e[2] += 1
            e = res["rounded"]
            e[0] = max(e[0], np.linalg.norm(rounded[:3] - r)); e[1] = max(e[1], np.linalg.norm(rounded[3:] - v)); e[2] += 1
for key, (er, ev, cnt) in res.items():
    print("%-8s max |dr| %.3f m   max |dv| %.3f mm/s   (%d points)" % (key, er * 1e3, ev * 1e6, cnt))
