#!/bin/bash
# GPU call: the full -m gpu tier, smoke, and the default bench line (with its `secondary` block) -> gpurun_out/$TAG
T=${TAG:-r3a}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -q $PYTEST_ARGS > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.log 2>&1
timeout 600 python bench.py $BENCH_ARGS > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
grep -E "passed|failed" gpurun_out/$T/pytest.log | tail -2; tail -1 gpurun_out/$T/smoke.log
python tools/show_bench.py gpurun_out/$T/bench.json
exit 0
