#!/bin/bash
mkdir -p gpurun_out/r2u
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/r2u/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2u/pytest.log
for args in "--layout time" "--layout time --stride-align 16" "--deep 1522 --layout time" "--layout time --pos-only"; do
  echo "== bench $args" >> gpurun_out/r2u/bench.log
  timeout 120 python bench.py --no-cpu-baseline $args >> gpurun_out/r2u/bench.log 2>&1
done
