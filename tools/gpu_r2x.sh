#!/bin/bash
mkdir -p gpurun_out/r2x
python tools/sweep.py run > gpurun_out/r2x/sweep.log 2>&1
python tools/sweep.py run --sats 13478 --times 10000 --steps 100 --warmup 30 > gpurun_out/r2x/sweep10k.log 2>&1
