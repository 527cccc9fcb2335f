#!/bin/bash
mkdir -p gpurun_out/r2s
export TMPDIR=/tmp; R=$PWD
for v in base d2 d4; do
  cd /tmp
  ASTROZ_AMD_LIB=$R/tools/variants/lib_$v.so timeout 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2s/tr_$v -o t -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --deep 1522 > $R/gpurun_out/r2s/tr_$v.log 2>&1
  cd $R
done
python - <<'PY'
import sqlite3, glob
for v in ("base","d2","d4"):
    for f in glob.glob("gpurun_out/r2s/tr_%s/*.db"%v):
        print("==",v)
        for r in sqlite3.connect(f).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if r[4] > 1: print("%-70s calls=%-5d avg_us=%-9.2f pct=%.2f"%(r[0][:70],r[1],r[3]/1000.0 if r[3]>1e4 else r[3],r[4]))
import json
for v in ("base","d2","d4"):
    for ln in open("gpurun_out/r2s/tr_%s.log"%v):
        if ln.startswith("{"):
            j=json.loads(ln); print(v, "ms/step %.4f"%j["ms_per_step"])
PY
