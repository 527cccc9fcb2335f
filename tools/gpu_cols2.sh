#!/bin/bash
# kernel-level timings of the cols path (rocprofv3 kernel trace per variant)
T=${TAG:-cols2}
mkdir -p gpurun_out/$T
cd /tmp; export TMPDIR=/tmp
for f in $GRAFT_REPO_ROOT/tools/variants/lib_*.so; do
  n=$(basename $f .so)
  for args in "--layout time" ${EXTRA_CASES:+"$EXTRA_CASES"}; do
    ASTROZ_AMD_COLS=1 ASTROZ_AMD_LIB=$f rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary $args --steps 100 --warmup 30 > /tmp/out_$n.txt 2>&1
    echo "== $n $args: $(tail -1 /tmp/out_$n.txt | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.4f ms" % j["ms_per_step"])')" | tee -a $GRAFT_REPO_ROOT/gpurun_out/$T/kt.log
    python - <<PY | tee -a $GRAFT_REPO_ROOT/gpurun_out/$T/kt.log
import glob, sqlite3
for f in glob.glob("/tmp/prof_$n/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    for nm, calls, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if calls >= 50: print("   %-70s calls=%-5d avg_us=%.2f" % (nm[:70], calls, avg))
PY
    rm -rf /tmp/prof_$n
  done
done
exit 0
