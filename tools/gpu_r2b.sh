#!/bin/bash
mkdir -p gpurun_out/r2b
python tools/sweep.py run > gpurun_out/r2b/sweep_pv.log 2>&1
python tools/sweep.py run --pos-only > gpurun_out/r2b/sweep_p.log 2>&1
bash tools/profile.sh r2b > gpurun_out/r2b/profile.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r2b gpurun_out/r2b/summary.txt "fast path default" >> gpurun_out/r2b/profile.log 2>&1
