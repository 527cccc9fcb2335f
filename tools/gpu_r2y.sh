#!/bin/bash
mkdir -p gpurun_out/r2y
timeout 400 python -m pytest tests -m gpu -q -x > gpurun_out/r2y/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2y/pytest.log
python tools/sweep.py run > gpurun_out/r2y/sweep.log 2>&1
python tools/sweep.py run --sats 13478 --times 10000 --steps 100 --warmup 30 > gpurun_out/r2y/sweep10k.log 2>&1
python tools/sweep.py run --deep 1522 > gpurun_out/r2y/sweep_c3.log 2>&1
