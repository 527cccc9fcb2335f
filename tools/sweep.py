#!/usr/bin/env python3
"""Kernel tuning sweep (development tool).
  python tools/sweep.py build name1:-DX=1,-DY=2 name2:...   compile library variants here (hipcc
                                                            cross-compiles) into tools/variants/
  python tools/sweep.py run [bench.py args]                 on the GPU box: bench every variant
Variants of the SAME library, selected at run time with ASTROZ_AMD_LIB."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "tools", "variants")


def build(specs):
    os.makedirs(VAR, exist_ok=True)
    for f in os.listdir(VAR):
        os.remove(os.path.join(VAR, f))
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(VAR, "lib_%s.so" % name)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread"] + [f for f in flags.split(",") if f] + [
            "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "include"), "-o", out,
            os.path.join(ROOT, "astroz_amd/csrc/astroz_hip.hip"), os.path.join(ROOT, "astroz_amd/csrc/tle_host.cpp"),
            "-x", "none", os.path.join(ROOT, "astroz_amd/host_step.o")]  # (built by __graft_entry__.build())
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for name, pr in procs:
        _, err = pr.communicate()
        info, cur = [], None
        for ln in err.splitlines():
            if "Function Name" in ln:
                cur = ln.split("Function Name:")[1].split()[0]
            if cur and ("k_rowsILb1ELb0ELi0" in cur or "k_propagateILi1ELb1ELb0ELb0ELb0" in cur or "k_cols_fastILb1ELi0ELi0" in cur) and any(
                    k in ln for k in (" VGPRs:", "ScratchSize", "Occupancy")):
                info.append(("rows " if "k_rows" in cur else ("cols " if "k_cols" in cur else "prop ")) + ln.split("remark:")[1].strip().split(" [")[0])
        print(name, pr.returncode, "; ".join(info), flush=True)


def run(extra):
    for f in sorted(os.listdir(VAR)):
        if not f.endswith(".so"):
            continue
        name = f[4:-3]
        env = dict(os.environ, ASTROZ_AMD_LIB=os.path.join(VAR, f))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            pw = j.get("power") or {}
            print("%-14s %7.2f Gprops/s  launch %.4f ms   %s W  %s MHz" % (name, j["value"] / 1e9, j["roofline"]["avg_launch_ms"],
                  pw.get("socket_w_median"), pw.get("sclk_mhz_median")), flush=True)
        except Exception:
            print(name, "FAILED", r.stderr[-300:], flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
