#!/usr/bin/env python3
"""Kernel tuning sweep (development tool).  `build` compiles library variants here (hipcc
cross-compiles); `run` (on the GPU box) benches each variant x time-tile and prints a table."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "tools", "variants")
VARIANTS = {"full": (256, 1, 0), "nostore": (256, 1, 1), "nocompute": (256, 1, 2)}

def build():
    os.makedirs(VAR, exist_ok=True)
    for name, (blk, w, abl) in VARIANTS.items():
        out = os.path.join(VAR, "lib_%s.so" % name)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DAZ_BLOCK=%d" % blk,
               "-DAZ_MIN_WAVES=%d" % w, "-DAZ_ABLATE=%d" % abl, "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "include"), "-o", out,
               os.path.join(ROOT, "astroz_amd/csrc/astroz_hip.hip"), os.path.join(ROOT, "astroz_amd/csrc/tle_host.cpp")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        info = []
        cur = None
        for ln in r.stderr.splitlines():
            if "Function Name" in ln: cur = ln.split("Function Name:")[1].split()[0]
            if cur and "k_rowsILb1ELb0" in cur and any(k in ln for k in ("VGPRs:", "ScratchSize", "Occupancy")):
                info.append(ln.split("remark:")[1].strip().split(" [")[0])
        print(name, r.returncode, "; ".join(info))

def run(extra):
    rows = []
    tiles = [0]
    for name in VARIANTS:
        lib = os.path.join(VAR, "lib_%s.so" % name)
        if not os.path.exists(lib): continue
        for tile in tiles:
            env = dict(os.environ, ASTROZ_AMD_LIB=lib)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--no-cpu-baseline", "--tile", str(tile)] + extra
            r = subprocess.run(cmd, capture_output=True, text=True, env=env)
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                rows.append((name, tile, j["value"] / 1e9, j["roofline"]["avg_launch_ms"]))
                print("%-8s tile=%3d  %7.2f Gprops/s  launch %.4f ms" % rows[-1], flush=True)
            except Exception as e:
                print(name, tile, "FAILED", r.stderr[-300:], flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "build": build()
    else: run(sys.argv[2:])
