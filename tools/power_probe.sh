#!/bin/bash
# clocks and power while the row kernels run: is "arithmetic + stores" slower than either alone because the chip
# lowers its clock under the combined load?  Variants built by tools/sweep.py (AZ_ABLATE=1: arithmetic only, =2: stores only).
out=gpurun_out/power; mkdir -p $out
poll() { while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' ' >> $1; echo >> $1; sleep 0.15; done; }
for f in tools/variants/lib_*.so; do
  name=$(basename $f .so)
  for mode in "" "--f32-out"; do
    tag="${name}${mode// /_}"
    poll $out/$tag.smi & pp=$!
    ASTROZ_AMD_LIB=$f timeout 120 python bench.py --no-cpu-baseline --sats 13478 --times 10000 --steps 5000 --warmup 200 $mode > $out/$tag.json 2> $out/$tag.err
    kill $pp; wait $pp 2>/dev/null
  done
done
/opt/rocm/bin/rocm-smi --showclocks --showpower --csv > $out/header.txt 2>&1
