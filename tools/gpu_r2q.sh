#!/bin/bash
mkdir -p gpurun_out/r2q
timeout 120 python tools/sweep.py run --layout time --stride-align 16 > gpurun_out/r2q/sweep_tm_aligned.log 2>&1
