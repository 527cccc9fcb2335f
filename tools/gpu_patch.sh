#!/bin/bash
T=${TAG:-patch}
mkdir -p gpurun_out/$T
for g in 0 1 0 1; do
  ASTROZ_AMD_DEEP_PATCH=$g timeout 300 python bench.py --deep 1522 --layout time --steps 200 --warmup 50 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/g.json
  python - <<PY | tee -a gpurun_out/$T/ab.log
import json
j=json.loads(open("/tmp/g.json").read())
print("config 3 time-major  deep_patch=$g  ms_per_step %.4f  parity %s" % (j["ms_per_step"], j.get("parity")))
PY
done
ASTROZ_AMD_DEEP_PATCH=1 timeout 600 python -m pytest tests -m gpu -q -x -k "full_size or tile or time_major or layout" 2>&1 | tail -3
exit 0
