#!/bin/bash
mkdir -p gpurun_out/r2f
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
for args in "" "--no-fast-path" "--layout time" "--layout time --stride-align 16" "--deep 1522" "--deep 1522 --layout time" "--pos-only" "--f32-out" "--times 10000 --sats 13478"; do
  echo "== bench $args" >> gpurun_out/r2f/bench.log
  timeout 300 python bench.py --no-cpu-baseline $args >> gpurun_out/r2f/bench.log 2>&1
done
timeout 300 python bench.py > gpurun_out/r2f/bench_full.log 2>&1
