#!/usr/bin/env python3
"""profiles/r02_config2_sat_major_rocprofv3.txt (tools/profile.sh + tools/summarize_profile.py) -> profiles/latest_pmc.json,
stamped with the fingerprint of the kernel sources it was measured on (bench.py reports `traffic` only on a match)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
txt = open(os.path.join(ROOT, "profiles", "r02_config2_sat_major_rocprofv3.txt")).read()
W = F = 0.0
kern = []
for blk in txt.split("kernel: ")[1:]:
    name = blk.split("\n")[0].strip()
    w = re.search(r"WRITE_SIZE\s+per_dispatch=([\d.e+]+)", blk)
    f = re.search(r"FETCH_SIZE\s+per_dispatch=([\d.e+]+)", blk)
    if w and f:
        W += float(w.group(1)); F += float(f.group(1)); kern.append(name.replace("void ", "").replace("(PropArgs)", ""))
d = {"source": "profiles/r02_config2_sat_major_rocprofv3.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of "
               "`python bench.py --steps 3 --warmup 1 --precondition-ms 0 --no-cpu-baseline`)",
     "kernels": kern,
     "workload": "13478 sats x 1440 times fp64 pos+vel satellite-major (one step = k_rows_fast near-circular + k_rows_fast eccentric + k_rows redo pass)",
     "WRITE_SIZE_KB": W, "FETCH_SIZE_KB": F,
     "note": "bytes = (WRITE_SIZE + 2*FETCH_SIZE) * 1024 summed over the step's three kernels: FETCH_SIZE under-reports wide coalesced reads by 2x "
             "on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated there and taken at face value",
     "hbm_bytes_per_launch": (W + 2 * F) * 1024, "csrc_sha16": bench.csrc_fingerprint()}
json.dump(d, open(os.path.join(ROOT, "profiles", "latest_pmc.json"), "w"), indent=1)
print(d["hbm_bytes_per_launch"] / 1e6, "MB", d["csrc_sha16"])
