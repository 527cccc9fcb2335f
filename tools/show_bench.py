#!/usr/bin/env python3
"""print the headline and the secondary block of a bench.py JSON line: tools/show_bench.py gpurun_out/<tag>/bench.json"""
import json, sys
ls = list(open(sys.argv[1]))
fl = [l for l in ls if l.startswith("BENCH_FULL ")]
j = json.loads(fl[-1][11:]) if fl else json.loads([l for l in ls if l.startswith("{")][-1])
print("last line bytes:", len(ls[-1]))
print("headline ms/step %.4f  value %.4g  frac %.3f  parity %s  power %s" % (j["ms_per_step"], j["value"], j["roofline"]["frac"], j.get("parity"),
      {k: v for k, v in (j.get("power") or {}).items() if k in ("socket_w_median", "sclk_mhz_median")}))
if j.get("cpu_baseline"): print("cpu_baseline", j["cpu_baseline"].get("value"), j["cpu_baseline"].get("cores"))
for e in j.get("secondary", []):
    print("%-28s ms %-8s frac %-6s %s %s %s" % (e.get("key"), "%.4f" % e["ms_per_step"] if "ms_per_step" in e else None,
          "%.3f" % e["roofline"]["frac"] if "roofline" in e else None, e.get("parity"), e.get("cold_grid_call_ms", e.get("reader_1M", "")), e.get("failed", "")))
