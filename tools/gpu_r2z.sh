#!/bin/bash
mkdir -p gpurun_out/r2z
timeout 500 python -m pytest tests -m gpu -q -x > gpurun_out/r2z/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2z/pytest.log
for args in "" "--sats 13478 --times 10000 --steps 100 --warmup 30" "--config5-share" "--deep 1522" "--f32-out" "--layout time"; do
  echo "== bench $args" >> gpurun_out/r2z/bench.log
  timeout 200 python bench.py --no-cpu-baseline $args >> gpurun_out/r2z/bench.log 2>&1
done
