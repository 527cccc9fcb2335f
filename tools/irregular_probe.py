"""Irregular-grid time-major and satellite-major timings of one library build (ASTROZ_AMD_LIB), config 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from astroz_amd import _native, synth
pairs = synth.synth_catalog(13478, 0)
dev = _native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
dev.set_timing(False)
n, T = dev.n, 1440
times = np.arange(float(T)) + np.random.default_rng(7).uniform(-1 / 3, 1 / 3, T)
off = (synth.START_JD - dev.epochs) * 1440.0
st = torch.cuda.Stream(); torch.cuda.set_stream(st); sp = st.cuda_stream


mask = (np.random.default_rng(3).uniform(size=n) > 0.15).astype(np.uint8)


def run(layout, tile_kernel, label, m=None, tt=None):
    global times
    if tt is not None:
        times = tt
    shape = (T, n, 3) if layout == _native.TIME_MAJOR else (n, T, 3)
    pos = torch.empty(shape, dtype=torch.float64, device="cuda"); vel = torch.empty_like(pos)
    dev.set_tile_kernel(tile_kernel)
    dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=layout, stream=sp, mask=m)
    torch.cuda.synchronize()
    path = dev.last_path()
    for _ in range(300):
        dev.propagate_device_cached(pos.data_ptr(), vel.data_ptr(), layout=layout, stream=sp)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(100):
        dev.propagate_device_cached(pos.data_ptr(), vel.data_ptr(), layout=layout, stream=sp)
    e1.record(st); torch.cuda.synchronize()
    print("%-40s path=%3d  %.4f ms" % (label, path, e0.elapsed_time(e1) / 100), flush=True)


run(_native.TIME_MAJOR, 0, "irregular time-major, k_propagate")
run(_native.TIME_MAJOR, 1, "irregular time-major, default routing")
run(_native.SAT_MAJOR, 1, "irregular sat-major, k_rows")
run(_native.TIME_MAJOR, 1, "irregular masked time-major, k_tiles", mask)
run(_native.TIME_MAJOR, 0, "irregular masked time-major, k_propagate", mask)
tu = np.arange(float(T))
run(_native.TIME_MAJOR, 1, "uniform masked time-major, k_tiles", mask, tu)
run(_native.TIME_MAJOR, 0, "uniform masked time-major, k_propagate", mask, tu)
run(_native.TIME_MAJOR, 1, "uniform time-major, k_tiles_fast", None, tu)
