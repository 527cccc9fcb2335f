#!/bin/bash
# round 3, GPU call A: the full -m gpu tier, smoke, and the default bench line (with its `secondary` block)
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3a/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/pytest.log; tail -2 gpurun_out/r3a/smoke.log; python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r3a/bench.json") if l.startswith("{")][-1])
print("headline ms/step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"], "power", j.get("power"))
for e in j.get("secondary", []):
    print(e.get("key"), e.get("ms_per_step"), e.get("roofline", {}).get("frac"), e.get("parity"), e.get("cold_grid_call_ms"), e.get("failed"))
PY
