#!/bin/bash
mkdir -p gpurun_out/r2o
python tools/sweep.py run --f32-out --times 10000 > gpurun_out/r2o/sweep_f32.log 2>&1
python tools/sweep.py run --times 10000 > gpurun_out/r2o/sweep_f64.log 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2o/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --f32-out --times 10000 > $R/gpurun_out/r2o/trace.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for f in glob.glob("gpurun_out/r2o/trace/*.db"):
    for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-90s calls=%-5d avg_us=%-9.2f pct=%.2f"%(r[0][:90],r[1],r[3]/1000.0 if r[3]>1e4 else r[3],r[4]))
PY
