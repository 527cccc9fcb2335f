"""Timeline of one step's launches from a rocprofv3 kernel trace (development probe): gaps between consecutive bulk kernels and
where the eccentric / redo launches end relative to the bulk.  usage: timeline_probe.py <trace dir>"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
bulk = [r for r in rows if r[2].startswith("void k_rows_fast<true, 0, 0, false") or r[2].startswith("k_rows_fast<true, 0, 0, false")]
ecc = [r for r in rows if "k_rows_fast<true, 0, 0, true" in r[2]]
redo = [r for r in rows if "k_rows<true, false, 0, true>" in r[2]]
print("kernels", len(rows), "bulk", len(bulk), "ecc", len(ecc), "redo", len(redo))
if len(bulk) > 12:
    b = bulk[5:]
    gaps = np.array([b[i + 1][0] - b[i][1] for i in range(len(b) - 1)]) / 1e3
    period = np.array([b[i + 1][0] - b[i][0] for i in range(len(b) - 1)]) / 1e3
    dur = np.array([x[1] - x[0] for x in b]) / 1e3
    print("bulk duration us: median %.1f  period (start to start) %.1f  gap end->next start %.1f (min %.1f max %.1f)" % (
        np.median(dur), np.median(period), np.median(gaps), gaps.min(), gaps.max()))
    # for each bulk, the ecc / redo kernel that started inside its period
    def rel(lst, name):
        out_s, out_e = [], []
        j = 0
        for i in range(len(b) - 1):
            for x in lst:
                if b[i][0] - 20000 <= x[0] < b[i + 1][0] - 20000:
                    out_s.append((x[0] - b[i][0]) / 1e3)
                    out_e.append((x[1] - b[i][1]) / 1e3)
                    break
        if out_s:
            print("%s: start - bulk start %.1f us, end - bulk end %.1f us (median; max %.1f)" % (name, np.median(out_s), np.median(out_e), np.max(out_e)))
    rel(ecc, "ecc")
    rel(redo, "redo")
