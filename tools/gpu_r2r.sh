#!/bin/bash
mkdir -p gpurun_out/r2r
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/r2r/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2r/pytest.log
for args in "" "--deep 1522" "--f32-out --times 10000" "--times 10000" "--layout time"; do
  echo "== bench $args" >> gpurun_out/r2r/bench.log
  timeout 120 python bench.py --no-cpu-baseline $args >> gpurun_out/r2r/bench.log 2>&1
done
echo "== config5 share" >> gpurun_out/r2r/bench.log
timeout 300 python bench.py --config5-share >> gpurun_out/r2r/bench.log 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2r/trace_sat -o t -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r2r/trace_sat.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2r/trace_f32 -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --f32-out --times 10000 > $R/gpurun_out/r2r/trace_f32.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for d in ("trace_sat","trace_f32"):
    for f in glob.glob("gpurun_out/r2r/%s/*.db"%d):
        print("==",d)
        for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if r[4] > 0.5: print("%-90s calls=%-5d avg_us=%-9.2f pct=%.2f"%(r[0][:90],r[1],r[3]/1000.0 if r[3]>1e4 else r[3],r[4]))
PY
