#!/bin/bash
mkdir -p gpurun_out/r2m
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2m/pytest.log
bash tools/profile.sh r2m > gpurun_out/r2m/profile.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r2m gpurun_out/r2m/summary.txt "python bench.py --steps 100 --warmup 20 --no-cpu-baseline (default: config 2, satellite-major, fp64 pos+vel); PMC passes --steps 3 --warmup 1 --precondition-ms 0" >> gpurun_out/r2m/profile.log 2>&1
timeout 300 python bench.py > gpurun_out/r2m/bench_full.log 2>&1
