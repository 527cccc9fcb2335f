#!/usr/bin/env python3
"""print the top_kernels table of a rocprofv3 --kernel-trace --stats output directory: tools/trace_top.py <dir> [name filters...]"""
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    for nm, calls, tot, avg, pct in sqlite3.connect(f).cursor().execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if len(sys.argv) < 3 or any(k in nm for k in sys.argv[2:]):
            print("%-60s calls=%-5d avg_us=%-9.2f pct=%.2f" % (nm[:60], calls, avg, pct))
