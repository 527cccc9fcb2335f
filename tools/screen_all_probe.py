"""the all-vs-all screen (azh_screen_all_host: propagate + cell list on the GPU, config 2 x 120 steps at 10 km) in a loop: wall clock
per call; run under rocprofv3 --kernel-trace / --pmc (tools/profile_run.py --script) to see where the 1.8 ms go"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from astroz_amd import _native, synth
pairs = synth.synth_catalog(13478, 0)
dev = _native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
dev.set_timing(False)
times = np.arange(120.0)
off = (synth.START_JD - dev.epochs) * 1440.0
n_calls = int(sys.argv[sys.argv.index("--calls") + 1]) if "--calls" in sys.argv else 40
ws = []
for _ in range(n_calls + 2):
    t0 = time.perf_counter(); pr, tt = dev.screen_all(times, 10.0, off); ws.append((time.perf_counter() - t0) * 1e3)
ws = ws[2:]
print("azh_screen_all_host, 13,478 x 120 steps, 10 km: %.3f ms wall per call (median of %d; min %.3f), %d pairs" % (sorted(ws)[len(ws) // 2], len(ws), min(ws), len(tt)))
