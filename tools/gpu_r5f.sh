#!/bin/bash
T=${TAG:-r5f}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -q -x -k "window or shard or config4 or group or round5" > gpurun_out/$T/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest.log
tail -4 gpurun_out/$T/pytest.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for ch in 4 1 0; do
  timeout 300 python bench.py --force-sharded --chunks $ch --steps 50 --warmup 20 --no-cpu-baseline 2>gpurun_out/$T/fs_$ch.err | tail -1 > gpurun_out/$T/fs_$ch.json
  python - <<PY
import json
j=json.loads(open("gpurun_out/$T/fs_$ch.json").read())
c=j["config"]
print("chunks arg $ch ->", c.get("chunks"), "t_kernel_ms %.4f t_total_ms %.4f t_replicate_ms %.4f  ms_per_step %.4f" % (c["t_kernel_ms"], c["t_total_ms"], c["t_replicate_ms"], j["ms_per_step"]), j.get("parity"))
PY
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_fs -o t -- python $GRAFT_REPO_ROOT/bench.py --force-sharded --chunks 4 --no-gather --steps 50 --warmup 20 --no-cpu-baseline > /tmp/fs_trace.txt 2>&1
python - <<PY | tee $GRAFT_REPO_ROOT/gpurun_out/$T/fs_trace.txt
import glob, sqlite3
for f in glob.glob("/tmp/prof_fs/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    for nm, calls, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if calls >= 20: print("   %-70s calls=%-5d avg_us=%.2f" % (nm[:70], calls, avg))
PY
exit 0
