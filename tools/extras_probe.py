"""Timings of the paths next to the headline kernel: fp32 outputs,
fused single-target screen, all-vs-all coarse screen, one satellite x many times, catalog ingest.
Run on the GPU box:  python tools/extras_probe.py  (prints one JSON object)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
import torch  # noqa: E402

from astroz_amd import _native, synth  # noqa: E402

out = {}
pairs = synth.synth_catalog(13478, 0)
t0 = time.perf_counter()
dev = _native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
out["ingest_13478_tle_lines_s"] = time.perf_counter() - t0
text = synth.pairs_to_text(pairs)
t0 = time.perf_counter()
dev2 = _native.DeviceConstellation.from_tle_text(text, 1, 0)
out["ingest_13478_tle_text_s"] = time.perf_counter() - t0
dev2.close()
n, nt = dev.n, 1440
times = np.arange(nt, dtype=np.float64)
off = (synth.START_JD - dev.epochs) * 1440.0
st = torch.cuda.Stream()


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    dev.synchronize(); st.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dev.synchronize(); st.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


dev.set_timing(False)
# fp32 outputs
for vel in (True, False):
    p32 = torch.empty((n, nt, 3), dtype=torch.float32, device="cuda")
    v32 = torch.empty_like(p32) if vel else None
    dev.propagate_device(times, off, p32.data_ptr(), v32.data_ptr() if vel else None, layout=_native.SAT_MAJOR,
                         stream=st.cuda_stream, f32=True)
    ms = timed(lambda: dev.propagate_device_cached(p32.data_ptr(), v32.data_ptr() if vel else None,
                                                   layout=_native.SAT_MAJOR, stream=st.cuda_stream, f32=True))
    out["f32_sat_major_%s_ms" % ("posvel" if vel else "pos")] = ms
    out["f32_sat_major_%s_gprops" % ("posvel" if vel else "pos")] = n * nt / ms / 1e6
# fused single-target screen (host call: includes the 160-KB result copy)
dev.set_timing(True)
for _ in range(3):
    d, ti = dev.screen_target(times, 0, 10.0, off)
out["screen_target_kernel_ms"] = dev.last_kernel_ms()
t0 = time.perf_counter()
for _ in range(10):
    d, ti = dev.screen_target(times, 0, 10.0, off)
out["screen_target_call_ms"] = (time.perf_counter() - t0) / 10 * 1e3
out["screen_target_gprops_kernel"] = n * nt / out["screen_target_kernel_ms"] / 1e6
out["screen_target_hits_10km"] = int((d < 10.0).sum())
# all-vs-all, fused with the propagation
for thr in (10.0, 50.0):
    dev.screen_all(times[:64], thr, off)
    t0 = time.perf_counter()
    pp, tt = dev.screen_all(times, thr, off)
    out["screen_all_%gkm_s" % thr] = time.perf_counter() - t0
    out["screen_all_%gkm_pairs" % thr] = int(len(tt))
# one satellite x many times (Satrec.sgp4_array / sgp4_propagate_batch), host pointers
for m in (1_000_000, 10_000_000):
    ts = np.linspace(0.0, 14400.0, m)
    dev.propagate_one(0, ts[:1000])
    t0 = time.perf_counter()
    e, r, v = dev.propagate_one(0, ts)
    dt = time.perf_counter() - t0
    out["one_sat_%d_times_host_s" % m] = dt
    out["one_sat_%d_mprops_host" % m] = m / dt / 1e6
# scalar calls: python-sgp4 style Satrec.sgp4(jd, fr) in a loop, and the raw C-ABI call
from astroz_amd.api import Satrec, WGS72  # noqa: E402
sat = Satrec.twoline2rv(pairs[0][0], pairs[0][1], WGS72)
sat.sgp4(sat.jdsatepoch, sat.jdsatepochF)
t0 = time.perf_counter()
for k in range(500):
    sat.sgp4(sat.jdsatepoch, sat.jdsatepochF + k / 1440.0)
out["Satrec_sgp4_scalar_call_us"] = (time.perf_counter() - t0) / 500 * 1e6
one = np.array([12.5])
dev.propagate_one(0, one)
t0 = time.perf_counter()
for k in range(500):
    dev.propagate_one(0, one)
out["propagate_one_scalar_call_us"] = (time.perf_counter() - t0) / 500 * 1e6
print(json.dumps(out, indent=1))
