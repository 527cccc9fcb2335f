#!/bin/bash
mkdir -p gpurun_out/r2l
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
for args in "" "--tile 1440" "--tile 512" "--tile 384" "--tile 256" "--deep 1522" "--times 10000" "--sats 125000"; do
  echo "== bench $args" >> gpurun_out/r2l/bench.log
  timeout 300 python bench.py --no-cpu-baseline $args >> gpurun_out/r2l/bench.log 2>&1
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2l/trace_sat -o t -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r2l/trace_sat.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
for d in ("trace_sat",):
    for f in glob.glob("gpurun_out/r2l/%s/*.db"%d):
        print("==",d)
        for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%-90s calls=%-5d avg_us=%-9.2f pct=%.2f"%(r[0][:90],r[1],r[3]/1000.0 if r[3]>1e4 else r[3],r[4]))
PY
