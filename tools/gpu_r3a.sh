#!/bin/bash
mkdir -p gpurun_out/r3a
timeout 300 python -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "cached_replays" > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.log 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3a/trace -o trace -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r3a/trace.log 2>&1
