#!/bin/bash
mkdir -p gpurun_out/r2v
timeout 400 python -m pytest tests -m gpu -q -x > gpurun_out/r2v/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2v/pytest.log
for args in "--config5-share" "--f32-out" "--f32-out --sats 13478 --times 10000" ""; do
  echo "== bench $args" >> gpurun_out/r2v/bench.log
  timeout 200 python bench.py --no-cpu-baseline $args >> gpurun_out/r2v/bench.log 2>&1
done
