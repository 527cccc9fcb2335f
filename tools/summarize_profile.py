#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) outputs of tools/profile.sh  ->  a plain-text summary for profiles/.
usage: tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>.txt "<command description>" """
import glob, sqlite3, sys

src, dst, desc = sys.argv[1], sys.argv[2], sys.argv[3]
out = ["# rocprofv3 summary (" + desc + ")", "# source: " + src, ""]
out.append("## rocprofv3 --kernel-trace --stats  (top_kernels: name, calls, total_us, avg_us, pct)")
for f in sorted(glob.glob(src + "/trace/*.db")):
    cur = sqlite3.connect(f).cursor()
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        out.append("%-80s calls=%-4d total_us=%-12.3f avg_us=%-10.3f pct=%.2f" % (r[0][:80], r[1], r[2], r[3], r[4]))
    out.append("")
    out.append("## per-dispatch durations of the dominant kernel (us)")
    rows = [r for r in cur.execute("select name, (end-start)/1000.0 from kernels order by start")] if False else []
out.append("## rocprofv3 --pmc (separate passes; value = average per dispatch)")
vals = {}
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_grbm", "sq", "sq2", "fetch", "write", "grbm"):
    for f in sorted(glob.glob(src + "/%s/*.db" % d)):
        cur = sqlite3.connect(f).cursor()
        q = ("select kernel_name, counter_name, sum(value), count(*), avg(duration), max(vgpr_count), max(sgpr_count), "
             "max(lds_block_size), max(scratch_size), max(grid_size), max(workgroup_size) from counters_collection group by 1,2")
        for r in cur.execute(q):
            if "k_propagate" in r[0] or "k_rows" in r[0] or "k_tiles" in r[0]:
                vals[(r[0], r[1])] = r[2:]
kern = sorted(set(k for k, _ in vals))
for k in kern:
    first = [v for (kk, _), v in vals.items() if kk == k][0]
    out.append("kernel: %s" % k)
    out.append("  vgpr=%s sgpr=%s lds_bytes=%s scratch=%s grid=%s workgroup=%s" % first[3:])
    for (kk, c), v in sorted(vals.items()):
        if kk == k:
            out.append("  %-28s per_dispatch=%-14.6g dispatches=%-3d avg_dur_ns=%.0f" % (c, v[0] / v[1], v[1], v[2]))
    d = {c: v[0] / v[1] for (kk, c), v in vals.items() if kk == k}
    dur = {c: v[2] for (kk, c), v in vals.items() if kk == k}
    if "SQ_WAVES" in d and "SQ_INSTS_VALU" in d:
        out.append("  derived: VALU instr per wave = %.0f" % (d["SQ_INSTS_VALU"] / d["SQ_WAVES"]))
    if "WRITE_SIZE" in d:
        out.append("  derived: HBM write traffic = %.1f MB/dispatch (WRITE_SIZE KB x 1024)" % (d["WRITE_SIZE"] * 1024 / 1e6))
    if "FETCH_SIZE" in d:
        out.append("  derived: HBM read traffic  = %.1f MB/dispatch raw (FETCH_SIZE KB x 1024; x2 per MI355X_MICROARCH.md = %.1f MB)" %
                   (d["FETCH_SIZE"] * 1024 / 1e6, 2 * d["FETCH_SIZE"] * 1024 / 1e6))
    out.append("")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out))
