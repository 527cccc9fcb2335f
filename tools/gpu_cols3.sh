#!/bin/bash
T=${TAG:-cols7}
mkdir -p gpurun_out/$T
for args in "--layout time" "--layout time --stride-align 16"; do
  echo "== $args  (tiles)" | tee -a gpurun_out/$T/sweep.log
  ASTROZ_AMD_COLS=0 python tools/sweep.py run $args --steps 100 --warmup 30 2>&1 | tee -a gpurun_out/$T/sweep.log
  echo "== $args  (cols)" | tee -a gpurun_out/$T/sweep.log
  ASTROZ_AMD_COLS=1 python tools/sweep.py run $args --steps 100 --warmup 30 2>&1 | tee -a gpurun_out/$T/sweep.log
done
exit 0
