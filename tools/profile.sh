#!/bin/bash
# rocprofv3 passes for the headline bench (run on the GPU box from the repo root).
#   tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/...
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 20 --no-cpu-baseline $BENCH_EXTRA"
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
PM="python $REPO/bench.py --steps 3 --warmup 1 --precondition-ms 0 --no-cpu-baseline $BENCH_EXTRA"
timeout 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- $PM > $OUT/pmc_sq.log 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F64 -d $OUT/pmc_sq2 -o pmc -- $PM > $OUT/pmc_sq2.log 2>&1
timeout 180 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $PM > $OUT/pmc_fetch.log 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $PM > $OUT/pmc_write.log 2>&1
timeout 180 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_grbm -o pmc -- $PM > $OUT/pmc_grbm.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -40
python - <<PY
import csv, glob, collections
out = "$OUT"
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    print("==", f); print(open(f).read()[:3000])
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_grbm"):
    for f in sorted(glob.glob(out + "/%s/**/*counter_collection.csv" % d, recursive=True)):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "")[:60], row.get("Counter_Name"))
            agg[k][0] += float(row.get("Counter_Value", 0)); agg[k][1] += 1
        print("==", d)
        for (kn, cn), (v, n) in sorted(agg.items()):
            if "k_propagate" in kn or "k_rows" in kn: print("%-60s %-28s sum=%.6g dispatches=%d per_dispatch=%.6g" % (kn, cn, v, n, v / max(n, 1)))
PY
