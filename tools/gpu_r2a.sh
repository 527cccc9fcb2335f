#!/bin/bash
# round-2 first GPU pass: parity tests on the new fast path + A/B benches
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
for args in "" "--no-fast-path" "--layout time" "--layout time --no-fast-path" "--layout time --stride-align 16" "--deep 1522" "--deep 1522 --layout time" "--pos-only" "--f32-out"; do
  echo "== bench $args" >> gpurun_out/r2a/bench.log
  timeout 300 python bench.py --no-cpu-baseline $args >> gpurun_out/r2a/bench.log 2>&1
done
timeout 300 python bench.py > gpurun_out/r2a/bench_full.log 2>&1
