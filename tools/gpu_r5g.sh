#!/bin/bash
T=${TAG:-r5g}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q -x > gpurun_out/$T/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest.log
tail -4 gpurun_out/$T/pytest.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29661 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 300 python bench.py --force-sharded --chunks 4 --steps 100 --warmup 30 --no-cpu-baseline 2>gpurun_out/$T/fs.err | tail -1 > gpurun_out/$T/fs.json
python - <<PY
import json
j=json.loads(open("gpurun_out/$T/fs.json").read()); c=j["config"]
print({k:c.get(k) for k in ("chunks","t_kernel_ms","t_kernel_graphs_ms","t_total_ms","t_replicate_ms")}, j.get("parity"))
PY
exit 0
