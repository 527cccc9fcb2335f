import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle
from astroz_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' | head -5")
pairs = synth.synth_catalog(4096, 0, seed=3)
cat = oracle.Catalog.from_pairs(pairs, 1)
times = np.arange(1440.0); off = (synth.START_JD - cat.epoch_jd) * 1440
for lay in (oracle.SAT_MAJOR,):
    for th in (1, 8, 16, 32, 64, 128):
        out = cat.propagate(times, off, layout=lay, threads=th)
        t0 = time.perf_counter()
        for _ in range(3): cat.propagate(times, off, layout=lay, threads=th, out=out)
        dt = (time.perf_counter() - t0) / 3
        print(lay, th, "%.1f M props/s" % (4096 * 1440 / dt / 1e6))
