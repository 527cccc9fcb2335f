#!/bin/bash
# round-2 profile set (rocprofv3 kernel traces + FETCH/WRITE/SQ PMC passes), each pass under its own timeout
mkdir -p gpurun_out/prof_set
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/prof_set/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/prof_set/pytest.log
BENCH_EXTRA="" bash tools/profile.sh r02_sat > gpurun_out/prof_set/prof_sat.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r02_sat gpurun_out/prof_set/r02_config2_sat_major.txt "python bench.py --steps 100 --warmup 20 --no-cpu-baseline (default: config 2, satellite-major, fp64 pos+vel); PMC passes --steps 3 --warmup 1 --precondition-ms 0" > /dev/null 2>&1
BENCH_EXTRA="--layout time" bash tools/profile.sh r02_tm > gpurun_out/prof_set/prof_tm.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r02_tm gpurun_out/prof_set/r02_config2_time_major.txt "python bench.py --layout time ... (config 2, time-major, fp64 pos+vel)" > /dev/null 2>&1
BENCH_EXTRA="--deep 1522" bash tools/profile.sh r02_c3 > gpurun_out/prof_set/prof_c3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r02_c3 gpurun_out/prof_set/r02_config3_sat_major.txt "python bench.py --deep 1522 ... (config 3: 13,478 near-earth + 1,522 deep-space, satellite-major)" > /dev/null 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_set/trace_c5 -o t -- python $R/bench.py --config5-share --steps 10 --warmup 3 > $R/gpurun_out/prof_set/trace_c5.log 2>&1
cd $R
python - <<'PY' > gpurun_out/prof_set/r02_config5_share_trace.txt 2>&1
import sqlite3, glob
print("# rocprofv3 --kernel-trace --stats -- python bench.py --config5-share --steps 10 --warmup 3  (125,000 sats x 10,000 times, fp32 pos+vel, one GPU)")
for f in glob.glob("gpurun_out/prof_set/trace_c5/*.db"):
    for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-90s calls=%-5d total_us=%-12.1f avg_us=%-10.2f pct=%.2f"%(r[0][:90],r[1],r[2]/1000.0,r[3]/1000.0,r[4]))
PY
for args in "" "--layout time" "--deep 1522" "--deep 1522 --layout time" "--pos-only" "--f32-out" "--times 10000" "--sats 125000" "--no-fast-path"; do
  echo "== bench $args" >> gpurun_out/prof_set/bench.log
  timeout 120 python bench.py --no-cpu-baseline $args >> gpurun_out/prof_set/bench.log 2>&1
done
timeout 200 python bench.py > gpurun_out/prof_set/bench_full.log 2>&1
timeout 300 python bench.py --config5-share > gpurun_out/prof_set/bench_c5.log 2>&1
