#!/bin/bash
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2g/trace_sat -o t -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r2g/trace_sat.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2g/trace_tm -o t -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --layout time > $R/gpurun_out/r2g/trace_tm.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2g/trace_deep -o t -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --deep 1522 > $R/gpurun_out/r2g/trace_deep.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for d in ("trace_sat","trace_tm","trace_deep"):
    for f in glob.glob("gpurun_out/r2g/%s/*.db"%d):
        print("==",d)
        for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%-90s calls=%-5d avg_us=%-9.2f pct=%.2f"%(r[0][:90],r[1],r[3]/1000.0 if r[3]>1e4 else r[3],r[4]))
PY
