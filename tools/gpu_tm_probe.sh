#!/bin/bash
i=0
for e in "X=1" "AZ_SKIP_TRANSPOSE=1" "AZ_SKIP_DEEP=1" "AZ_SKIP_DEEP=1 AZ_SKIP_TRANSPOSE=1"; do
  i=$((i+1))
  env $e python tools/profile_run.py tmprobe_e$i --steps 50 -- --deep 1522 --layout time > /dev/null 2>&1
  echo "== $e"; sed -n 5,5p gpurun_out/profiles/tmprobe_e$i.txt; sed -n 8,11p gpurun_out/profiles/tmprobe_e$i.txt | cut -c1-30,86-140
done
