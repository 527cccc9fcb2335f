#!/bin/bash
# same-box A/B of the library variants under tools/variants on the cases given in CASES (semicolon-separated bench.py argument lists)
T=${TAG:-ab}
mkdir -p gpurun_out/$T
IFS=';' read -ra CS <<< "${CASES:---layout time;--layout time --deep 1522}"
for rep in 1 2; do
for args in "${CS[@]}"; do
  echo "== $args" | tee -a gpurun_out/$T/ab.log
  python tools/sweep.py run $args --steps 200 --warmup 50 2>&1 | tee -a gpurun_out/$T/ab.log
done
done
exit 0
