#!/bin/bash
T=${TAG:-r5b}
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -x > gpurun_out/$T/pytest_r5.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest_r5.log
tail -5 gpurun_out/$T/pytest_r5.log
ASTROZ_AMD_COLS=1 timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/$T/pytest_cols.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest_cols.log
tail -5 gpurun_out/$T/pytest_cols.log
exit 0
