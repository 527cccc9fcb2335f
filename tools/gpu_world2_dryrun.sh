#!/bin/bash
# round 6: the N > 1 code path of bench.py with a REAL world size on a one-GPU box: every rank on device 0, gloo process group
# (ASTROZ_BENCH_DRYRUN_ONE_DEVICE=1).  Checks that every rank reaches every collective -- timing reductions, the sharded screen's
# agreement + object gather, the per-rank certificates -- and that rank 0's line comes out; the timings mean nothing.
T=gpurun_out/dry
mkdir -p $T
export ASTROZ_BENCH_DRYRUN_ONE_DEVICE=1
for W in 2 3; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2956$W bench.py --gpus $W --steps 5 --warmup 2 --precondition-ms 0 --sats 3000 --times 300 --cpu-seconds 1 > $T/w$W.out 2> $T/w$W.err
  echo "world $W rc=$?"; tail -1 $T/w$W.out > $T/w${W}_line.json
  python - <<PY
import json
try:
    j = json.loads(open("$T/w${W}_line.json").read())
    c = j["config"]
    print({k: j.get(k) for k in ("n_gpus", "scaling", "ms_per_step", "data")})
    print("sharded_screen:", c.get("sharded_screen"))
    print("parity:", j.get("parity"))
    print("cpu_baseline:", (j.get("cpu_baseline") or {}).get("value"), "t_kernel", c.get("t_kernel_ms"), "replicate", c.get("t_replicate_ms"), "group_host", c.get("group_host"))
except Exception as e:
    print("no line:", e); print(open("$T/w$W.err").read()[-3000:])
PY
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --scaling weak --f32-out --sats 4000 --times 512 --steps 3 --warmup 1 --precondition-ms 0 --cpu-seconds 1 > $T/weak.out 2> $T/weak.err
echo "weak rc=$?"; tail -1 $T/weak.out | head -c 1500; echo
tail -5 $T/weak.err
