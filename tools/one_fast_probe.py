"""k_one_fast (azh_propagate_one_device, 10^7 and 10^6 device-resident times of one satellite) for one library build (ASTROZ_AMD_LIB)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astroz_amd import synth, _native
from oracle import oracle
pairs = synth.synth_catalog(2000, 0)
cuda = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); sp = st.cuda_stream
devs = _native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
iecc = int(np.argmax(devs.field("ecco")))
for name, idx in (("leo", 0), ("ecc", iecc)):
    dev = _native.DeviceConstellation.from_tle_lines([pairs[idx]], 1, 0)
    dev.set_timing(False)
    cat = oracle.Catalog.from_pairs([pairs[idx]], oracle.WGS72)
    for n, span in ((10_000_000, 14400.0), (1_048_576, 1440.0 * 7)):
        times = np.linspace(0.0, span, n)
        ts = torch.as_tensor(times, device=cuda)
        po = torch.empty((n, 3), dtype=torch.float64, device=cuda); ve = torch.empty_like(po)
        torch.cuda.synchronize()
        for _ in range(20):
            dev.propagate_one_device(0, ts.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sp)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(50):
            dev.propagate_one_device(0, ts.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sp)
        e1.record(st); torch.cuda.synchronize()
        pick = np.unique(np.linspace(0, n - 1, 4096).astype(np.int64))
        _, p0, v0 = cat.propagate(times[pick], None, layout=oracle.SAT_MAJOR)
        sel = torch.as_tensor(pick, device=cuda)
        print("%-4s n=%-9d %.4f ms  stats %s  dr %.2e dv %.2e" % (name, n, e0.elapsed_time(e1) / 50, dev.last_one_stats(),
              np.abs(po[sel].cpu().numpy() - p0[0]).max(), np.abs(ve[sel].cpu().numpy() - v0[0]).max()), flush=True)
        del ts, po, ve
