#!/bin/bash
mkdir -p gpurun_out/r2n
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2n/pytest.log
for args in "--f32-out" "--f32-out --f32-rounded" "--f32-out --pos-only" "--f32-out --times 10000"; do
  echo "== bench $args" >> gpurun_out/r2n/bench.log
  timeout 300 python bench.py --no-cpu-baseline $args >> gpurun_out/r2n/bench.log 2>&1
done
echo "== config5 share" >> gpurun_out/r2n/bench.log
timeout 900 python bench.py --config5-share >> gpurun_out/r2n/bench.log 2>&1
echo "== config5 share rounded" >> gpurun_out/r2n/bench.log
timeout 900 python bench.py --config5-share --f32-rounded >> gpurun_out/r2n/bench.log 2>&1
