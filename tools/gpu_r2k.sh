#!/bin/bash
mkdir -p gpurun_out/r2k
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log
python tools/sweep.py run > gpurun_out/r2k/sweep_sat.log 2>&1
for args in "--deep 1522" "--times 10000" "--pos-only" "--f32-out"; do
  echo "== bench $args" >> gpurun_out/r2k/bench.log
  timeout 300 python bench.py --no-cpu-baseline $args >> gpurun_out/r2k/bench.log 2>&1
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2k/trace_sat -o t -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r2k/trace_sat.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for d in ("trace_sat",):
    for f in glob.glob("gpurun_out/r2k/%s/*.db"%d):
        print("==",d)
        for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%-90s calls=%-5d avg_us=%-9.2f pct=%.2f"%(r[0][:90],r[1],r[3]/1000.0 if r[3]>1e4 else r[3],r[4]))
PY
