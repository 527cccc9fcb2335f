#!/bin/bash
mkdir -p gpurun_out/r2j
python tools/sweep.py run > gpurun_out/r2j/sweep_sat.log 2>&1
python tools/sweep.py run --layout time > gpurun_out/r2j/sweep_tm.log 2>&1
for args in "--layout time --tile 8" "--layout time --tile 12" "--layout time --tile 32" "--deep 1522" "--times 10000"; do
  echo "== bench $args" >> gpurun_out/r2j/bench.log
  timeout 300 python bench.py --no-cpu-baseline $args >> gpurun_out/r2j/bench.log 2>&1
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2j/trace_sat -o t -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r2j/trace_sat.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for d in ("trace_sat",):
    for f in glob.glob("gpurun_out/r2j/%s/*.db"%d):
        print("==",d)
        for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%-90s calls=%-5d avg_us=%-9.2f pct=%.2f"%(r[0][:90],r[1],r[3]/1000.0 if r[3]>1e4 else r[3],r[4]))
PY
