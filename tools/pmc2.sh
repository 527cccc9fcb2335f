#!/bin/bash
# memory-pipeline stall counters for the headline kernel (separate --pmc passes; run on the GPU box).
# Every pass runs under `timeout`: on this pool the TA_*_sum derived metrics made rocprofv3 abort
# (signal 6) and hang until the call's limit -- they are no longer collected.
OUT=$PWD/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
PM="python $REPO/bench.py --steps 3 --warmup 1 --precondition-ms 0 --no-cpu-baseline $BENCH_EXTRA"
i=0
for set in "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_SRC_FIFO_FULL_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "MemUnitStalled VALUBusy GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o pmc -- $PM > $OUT/p$i.log 2>&1 || echo "pass $i failed or timed out"
done
cd $REPO
python - <<PY
import glob, sqlite3
for f in sorted(glob.glob("$OUT/p*/*.db")):
    cur = sqlite3.connect(f).cursor()
    try:
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by 1,2"):
            if "k_rows" in r[0] or "k_propagate" in r[0]:
                print("%-40s %-45s per_dispatch=%-14.6g n=%d dur_ns=%.0f" % (r[0][:40], r[1], r[2], r[3], r[4]))
    except Exception as e:
        print(f, "ERR", e)
PY
