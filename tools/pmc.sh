#!/bin/bash
# quick PMC pass on a given library variant: tools/pmc.sh <tag> [lib.so]
TAG=$1; LIB=$2
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; R=$PWD
[ -n "$LIB" ] && export ASTROZ_AMD_LIB=$R/$LIB
cd /tmp
PM="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/sq -o p -- $PM > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $OUT/sq2 -o p -- $PM > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p -- $PM > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o p -- $PM > $OUT/w.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/grbm -o p -- $PM > $OUT/g.log 2>&1
cd $R
python - <<PY
import sqlite3, glob
out="$OUT"
for f in glob.glob(out+"/trace/*.db"):
    for r in sqlite3.connect(f).cursor().execute("select * from top_kernels"):
        if "k_prop" in r[0]: print("TRACE", r)
vals={}
for d in ("sq","sq2","fetch","write","grbm"):
    for f in glob.glob(out+"/%s/*.db"%d):
        for r in sqlite3.connect(f).cursor().execute("select kernel_name, counter_name, sum(value), count(*), avg(duration) from counters_collection group by 1,2"):
            if "k_propagate" in r[0]: vals[r[1]]=(r[2]/r[3], r[4])
for k,v in sorted(vals.items()): print("%-26s %.6g (dur %.0f ns)"%(k,v[0],v[1]))
PY
