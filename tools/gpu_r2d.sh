#!/bin/bash
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
python tools/sweep.py run > gpurun_out/r2d/sweep_pv.log 2>&1
python tools/sweep.py run --no-fast-path > gpurun_out/r2d/sweep_pv_generic.log 2>&1
python tools/clock_probe.py --seconds 1.0 --tag fast_lds > gpurun_out/r2d/probe.log 2>&1
