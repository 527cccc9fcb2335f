"""the fused single-target screen, repeated on device buffers (config 2): per-call time; run under rocprofv3 --kernel-trace for the launch list"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from astroz_amd import _native, synth
pairs = synth.synth_catalog(13478, 0)
dev = _native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
dev.set_timing(False)
times = np.arange(1440.0)
off = (synth.START_JD - dev.epochs) * 1440.0
st = torch.cuda.Stream(); torch.cuda.set_stream(st); sp = st.cuda_stream
d = torch.empty(dev.n, dtype=torch.float64, device="cuda"); ti = torch.empty(dev.n, dtype=torch.int32, device="cuda")
import ctypes as C
L = _native.lib()
def call():
    rc = L.azh_screen_target_device(dev._h, times.ctypes.data, len(times), off.ctypes.data, 5, C.c_double(2000.0), C.c_double(0.0), d.data_ptr(), ti.data_ptr(), sp)
    assert rc == 0, rc
import time
n_calls = int(sys.argv[sys.argv.index("--calls") + 1]) if "--calls" in sys.argv else 200
for _ in range(max(2, n_calls // 4)): call()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record(st)
for _ in range(n_calls): call()
e1.record(st); torch.cuda.synchronize()
print("fused screen, config 2 (13,478 x 1,440), device buffers: %.4f ms per call (HIP events), %.4f ms wall incl. the final synchronize" % (
    e0.elapsed_time(e1) / n_calls, (time.perf_counter() - t0) * 1e3 / n_calls))
# the host-returning call (what `screen(..., target=)` makes): staging + kernels + the D2H of the two result vectors
ws = []
for _ in range(max(3, n_calls // 10)):
    t0 = time.perf_counter(); dev.screen_target(times, 5, 2000.0, off); ws.append((time.perf_counter() - t0) * 1e3)
print("azh_screen_target_host: %.4f ms wall per call (median of %d)" % (sorted(ws)[len(ws) // 2], len(ws)))
