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
for _ in range(50): call()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(200): call()
e1.record(st); torch.cuda.synchronize()
print("fused screen: %.4f ms per call" % (e0.elapsed_time(e1) / 200))
