#!/bin/bash
mkdir -p gpurun_out/r2e
python tools/sweep.py run > gpurun_out/r2e/sweep_pv.log 2>&1
python tools/sweep.py run --pos-only > gpurun_out/r2e/sweep_p.log 2>&1
