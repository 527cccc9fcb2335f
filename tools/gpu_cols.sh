#!/bin/bash
# round 5: the lane = satellite time-major kernel (k_cols_fast) -- parity through the GPU tier with the path switched on, then
# same-box timings against the tile kernel for every library variant under tools/variants
T=${TAG:-cols}
mkdir -p gpurun_out/$T
if [ -z "$SKIP_TESTS" ]; then
ASTROZ_AMD_COLS=1 timeout 900 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > gpurun_out/$T/pytest_cols.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_cols.log
tail -15 gpurun_out/$T/pytest_cols.log
fi
for args in "--layout time" "--layout time --deep 1522" "--layout time --grid jdfr" ${EXTRA_CASES:+"$EXTRA_CASES"}; do
  echo "== $args  (tiles)" | tee -a gpurun_out/$T/sweep.log
  ASTROZ_AMD_COLS=0 python tools/sweep.py run $args --steps 100 --warmup 30 2>&1 | tee -a gpurun_out/$T/sweep.log
  echo "== $args  (cols)" | tee -a gpurun_out/$T/sweep.log
  ASTROZ_AMD_COLS=1 python tools/sweep.py run $args --steps 100 --warmup 30 2>&1 | tee -a gpurun_out/$T/sweep.log
done
exit 0
