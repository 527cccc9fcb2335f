"""Where the time of a host-returning call goes (run on the GPU box): fresh vs touched numpy arrays, transparent huge
pages, pageable vs pinned D2H, host memcpy rate.  Prints one JSON object."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import torch
from astroz_amd import synth, _native
from astroz_amd.api import SatrecArray, Satrec

out = {"thp": open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "cpus": os.cpu_count()}
libc = ctypes.CDLL("libc.so.6", use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def huge(a):
    p = a.ctypes.data
    lo = (p + 4095) & ~4095
    return libc.madvise(lo, a.nbytes - (lo - p), 14)


pairs = synth.synth_catalog(13478, 0)
n, T = 13478, 1440
arr = SatrecArray([Satrec.twoline2rv(a, b) for a, b in pairs])
jd = np.full(T, synth.START_JD)
fr = 0.32853009 + np.arange(T) / 1440.0
ws = []
for rep in range(4):
    t0 = time.perf_counter(); e, r, v = arr.sgp4(jd, fr); ws.append(time.perf_counter() - t0)
    del e, r, v
out["SatrecArray.sgp4_s"] = ws
dev = arr._dev
times = np.arange(float(T))
off = (synth.START_JD - dev.epochs) * 1440.0
GB = 2 * n * T * 24 / 1e9


def timed(pos, vel):
    t0 = time.perf_counter(); dev.propagate_host(times, off, pos=pos, vel=vel); return time.perf_counter() - t0


res = {}
pos = np.empty((T, n, 3)); vel = np.empty_like(pos)
res["fresh_s"] = timed(pos, vel)
res["touched_s"] = [timed(pos, vel) for _ in range(3)]
del pos, vel
pos = np.empty((T, n, 3)); vel = np.empty_like(pos)
res["madvise_rc"] = [huge(pos), huge(vel)]
res["fresh_hugepage_s"] = timed(pos, vel)
res["touched_hugepage_s"] = [timed(pos, vel) for _ in range(3)]
out["propagate_host"] = res
out["GB"] = GB
# raw copies
d = torch.empty((T, n, 3), dtype=torch.float64, device="cuda")
hp = torch.empty((T, n, 3), dtype=torch.float64, pin_memory=True)
hq = torch.empty((T, n, 3), dtype=torch.float64)
hq.zero_()
cp = {}
for name, h in (("pinned", hp), ("pageable_touched", hq)):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    cp[name] = {"s": ts, "GB_per_s": d.numel() * 8 / min(ts) / 1e9}
out["d2h"] = cp
# host memcpy: numpy single thread, and torch (intra-op threads)
a = hp.numpy(); b = np.empty_like(a); b[:] = 0
t0 = time.perf_counter(); b[:] = a; out["np_copy_GBps"] = a.nbytes / (time.perf_counter() - t0) / 1e9
tb = torch.from_numpy(b)
t0 = time.perf_counter(); tb.copy_(hp); out["torch_copy_GBps"] = a.nbytes / (time.perf_counter() - t0) / 1e9
# first-touch cost of a fresh array, 4-KB pages vs huge pages
c = np.empty((T, n, 3)); t0 = time.perf_counter(); c[:] = 0; out["first_touch_4k_GBps"] = c.nbytes / (time.perf_counter() - t0) / 1e9
del c
c = np.empty((T, n, 3)); huge(c); t0 = time.perf_counter(); c[:] = 0; out["first_touch_huge_GBps"] = c.nbytes / (time.perf_counter() - t0) / 1e9
print(json.dumps(out, indent=1))
