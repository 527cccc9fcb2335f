#!/bin/bash
# what the driver runs at round end, plus a sustained-regime bench
mkdir -p gpurun_out/verify
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/verify/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/verify/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/verify/smoke.log 2>&1
timeout 300 python bench.py > gpurun_out/verify/bench.json 2> gpurun_out/verify/bench.err
timeout 300 python bench.py --no-cpu-baseline --steps 20000 --warmup 2000 > gpurun_out/verify/bench_sustained.json 2> gpurun_out/verify/bench_sustained.err
