#!/usr/bin/env python3
"""rocprofv3 passes around one bench.py command, summarised on the GPU box at measurement time.

  tools/profile_run.py <name> [--pmc] [--steps K] -- <bench.py args>
  tools/profile_run.py <name> [--pmc] --script tools/<probe>.py -- <probe args>     (any script instead of bench.py: the screens)

Writes gpurun_out/profiles/<name>.txt (+ .json): the --kernel-trace --stats table (per kernel: calls, average, total,
share) and, with --pmc, the counter passes of the MI355X guide (separate passes, never combined with traces): SQ
instruction / wait counters, FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE -- per kernel and dispatch, with the derived
figures (VALU instructions per wave, HBM bytes per dispatch with the guide's corrections).  The JSON carries the
sha256 fingerprint of astroz_amd/csrc AS IT IS ON THE BOX WHEN THE MEASUREMENT RUNS (bench_common.csrc_fingerprint):
profiles/latest_pmc.json is a copy of such a file, never re-stamped afterwards."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PMC_SETS = {
    "sq": "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY",
    "sq2": "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F64",
    "lds": "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "grbm": "GRBM_GUI_ACTIVE GRBM_COUNT",
}
KERNEL_PREFIXES = ("k_propagate", "k_rows", "k_tiles", "k_cols", "k_one", "k_deep", "k_prep", "k_classify", "k_screen", "k_cells", "k_plan")


def short(name):
    return name.replace("void ", "").replace("(PropArgs)", "")


def main():
    argv = sys.argv[1:]
    split = argv.index("--")
    own, bench_args = argv[:split], argv[split + 1:]
    name = own[0]
    pmc = "--pmc" in own
    steps = own[own.index("--steps") + 1] if "--steps" in own else "100"
    out_dir = os.path.join(ROOT, "gpurun_out", "profiles")
    raw = os.path.join(ROOT, "gpurun_out", "prof_raw", name)
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(raw, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    script = own[own.index("--script") + 1] if "--script" in own else None
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary"] + bench_args
    cmd_txt = "python bench.py --no-cpu-baseline --no-secondary " + " ".join(bench_args)
    if script:
        bench = [sys.executable, os.path.join(ROOT, script)] + bench_args
        cmd_txt = "python %s %s" % (script, " ".join(bench_args))
    tail_args = [] if script else ["--steps", steps, "--warmup", "20"]
    pmc_args = ["--calls", "3"] if script else ["--steps", "3", "--warmup", "1", "--precondition-ms", "0"]
    lines = ["# rocprofv3 summary: " + name, "# command: %s %s" % (cmd_txt, " ".join(tail_args)), ""]
    js = {"name": name, "command": cmd_txt, "kernels": {}}
    import bench_common as bench_mod
    js["csrc_sha16"] = bench_mod.csrc_fingerprint()
    lines.append("# csrc_sha16 (sources on the box at measurement time): " + js["csrc_sha16"])
    # ---- kernel trace
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.join(raw, "trace"), "-o", "trace", "--"] + bench +
                       tail_args, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    bl = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if script:
        lines += ["# " + ln for ln in r.stdout.splitlines() if ln.strip()][-8:]
        bl = []
    if bl:
        j = json.loads(bl[-1])
        js["bench_under_trace"] = {"ms_per_step": j["ms_per_step"], "value": j["value"]}
        lines.append("# bench line under the tracer: ms_per_step=%.4f value=%.4g" % (j["ms_per_step"], j["value"]))
    lines += ["", "## rocprofv3 --kernel-trace --stats (durations in microseconds)"]
    for f in sorted(glob.glob(os.path.join(raw, "trace", "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(f).cursor()
        for nm, calls, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            # rocpd's top_kernels view already divides the nanosecond timestamps by 1000: microseconds
            lines.append("%-86s calls=%-5d avg_us=%-10.2f total_us=%-12.1f pct=%.2f" % (short(nm)[:86], calls, avg, tot, pct))
            js["kernels"].setdefault(short(nm), {})["trace"] = {"calls": calls, "avg_us": avg, "total_us": tot, "pct": pct}
    # ---- counters
    if pmc:
        lines += ["", "## rocprofv3 --pmc (separate passes of `%s`; per-dispatch averages)" % " ".join(pmc_args)]
        vals = {}
        for tag, counters in PMC_SETS.items():
            d = os.path.join(raw, "pmc_" + tag)
            try:
                subprocess.run(["rocprofv3", "--pmc"] + counters.split() + ["-d", d, "-o", "pmc", "--"] + bench +
                               pmc_args, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=300)
            except subprocess.TimeoutExpired:
                lines.append("# pass %s timed out" % tag)
                continue
            for f in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
                cur = sqlite3.connect(f).cursor()
                q = ("select kernel_name, counter_name, sum(value), count(*), avg(duration), max(vgpr_count), max(sgpr_count), "
                     "max(lds_block_size), max(scratch_size), max(grid_size), max(workgroup_size) from counters_collection group by 1,2")
                try:
                    rows = list(cur.execute(q))
                except sqlite3.Error as exc:
                    lines.append("# pass %s: %r" % (tag, exc))
                    continue
                for row in rows:
                    if any(p in row[0] for p in KERNEL_PREFIXES):
                        vals[(short(row[0]), row[1])] = row[2:]
        for k in sorted(set(k for k, _ in vals)):
            first = [v for (kk, _), v in vals.items() if kk == k][0]
            lines.append("kernel: %s" % k)
            # rocprofv3 reports the VGPR allocation of a wave64 kernel on gfx950 in units of TWO registers (k_rows_fast: 40 for
            # the 74 -> 80 registers the ISA metadata shows): print the allocated count and the occupancy it
            # allows (512 registers per SIMD lane, at most 8 waves)
            vg = int(first[3] or 0) * 2
            wps = min(8, 512 // vg) if vg else 8
            lines.append("  vgpr_allocated=%d (rocprofv3 vgpr_count %s x 2) waves_per_simd=%d sgpr=%s lds_bytes=%s scratch=%s grid=%s workgroup=%s" % (
                (vg, first[3], wps) + tuple(first[4:])))
            d = {}
            for (kk, c), v in sorted(vals.items()):
                if kk == k:
                    d[c] = v[0] / v[1]
                    lines.append("  %-28s per_dispatch=%-14.6g dispatches=%-3d avg_dur_us=%.2f" % (c, v[0] / v[1], v[1], v[2] / 1e3))
            kj = js["kernels"].setdefault(k, {})
            kj["pmc"] = d
            kj["regs"] = dict(zip(("vgpr_count_reported", "sgpr", "lds_bytes", "scratch", "grid", "workgroup"), first[3:]))
            kj["regs"].update({"vgpr_allocated": vg, "waves_per_simd": wps})
            if "SQ_WAVES" in d and "SQ_INSTS_VALU" in d and d["SQ_WAVES"]:
                lines.append("  derived: VALU instr per wave = %.0f" % (d["SQ_INSTS_VALU"] / d["SQ_WAVES"]))
            if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES"):
                lines.append("  derived: SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f" % (
                    d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]))
            if "WRITE_SIZE" in d:
                lines.append("  derived: HBM write traffic = %.1f MB/dispatch (WRITE_SIZE KB x 1024)" % (d["WRITE_SIZE"] * 1024 / 1e6))
            if "FETCH_SIZE" in d:
                lines.append("  derived: HBM read traffic  = %.1f MB/dispatch (2 x FETCH_SIZE KB x 1024: the guide's gfx950 correction)" % (
                    2 * d["FETCH_SIZE"] * 1024 / 1e6))
            lines.append("")
        step_k = [k for k in js["kernels"] if "pmc" in js["kernels"][k] and k.startswith(("k_rows", "k_tiles", "k_cols", "k_propagate") + (("k_screen", "k_cells") if script else ()))]
        W = sum(js["kernels"][k]["pmc"].get("WRITE_SIZE", 0.0) for k in step_k)
        F = sum(js["kernels"][k]["pmc"].get("FETCH_SIZE", 0.0) for k in step_k)
        V = sum(js["kernels"][k]["pmc"].get("SQ_INSTS_VALU", 0.0) for k in step_k)
        js["step"] = {"kernels": step_k, "WRITE_SIZE_KB": W, "FETCH_SIZE_KB": F, "hbm_bytes_per_launch": (W + 2 * F) * 1024,
                      "valu_wave_insts_per_launch": V,
                      "note": "one step = the kernels listed; bytes = (WRITE_SIZE + 2 FETCH_SIZE) x 1024 (FETCH_SIZE under-reports wide "
                              "coalesced reads by 2x on gfx950, MI355X_MICROARCH.md HBM section; WRITE_SIZE at face value)"}
        lines.append("step total: HBM bytes per launch = %.1f MB, VALU wave instructions per launch = %.4g" % ((W + 2 * F) * 1024 / 1e6, V))
    open(os.path.join(out_dir, name + ".txt"), "w").write("\n".join(lines) + "\n")
    json.dump(js, open(os.path.join(out_dir, name + ".json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
