#!/bin/bash
# GPU: bench every library variant under tools/variants (tools/sweep.py build ...), then tile-size points on the base build
mkdir -p gpurun_out/${TAG:-sweep}
python tools/sweep.py run $SWEEP_ARGS 2>&1 | tee gpurun_out/${TAG:-sweep}/sweep.log
for t in $TILES; do
  echo "== base --tile $t" | tee -a gpurun_out/${TAG:-sweep}/sweep.log
  ASTROZ_AMD_LIB=tools/variants/lib_base.so python bench.py --no-cpu-baseline --no-secondary --tile $t $SWEEP_ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms %.2f G/s' % (j['ms_per_step'], j['value']/1e9))" | tee -a gpurun_out/${TAG:-sweep}/sweep.log
done
