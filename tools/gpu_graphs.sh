#!/bin/bash
# round 5: hipGraph replay of the cached launch sets -- correctness (GPU tier with graphs on = the default) and same-box A/B
T=${TAG:-graphs}
mkdir -p gpurun_out/$T
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/$T/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest.log
tail -4 gpurun_out/$T/pytest.log
fi
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29651 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for args in "" "--deep 1522" "--layout time" "--deep 1522 --layout time" "--force-sharded --chunks 4 --no-gather" "--force-sharded --chunks 4"; do
  for g in 0 1 0 1; do
    ASTROZ_AMD_GRAPHS=$g timeout 300 python bench.py $args --steps 200 --warmup 50 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/g.json
    python - <<PY | tee -a gpurun_out/$T/ab.log
import json
j=json.loads(open("/tmp/g.json").read()); c=j["config"]
print("%-44s graphs=$g  ms_per_step %.4f  t_kernel %s" % ("$args", j["ms_per_step"], c.get("t_kernel_ms")))
PY
  done
done
exit 0
