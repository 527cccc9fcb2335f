#!/bin/bash
# round 6, final block on the final sources: GPU tier, smoke, the bench lines the driver and the judge read, the profile set
T=gpurun_out/r6d
mkdir -p $T gpurun_out/profiles
timeout 1500 python -m pytest tests -m gpu -q -x > $T/pytest.log 2>&1; echo "rc=$?" >> $T/pytest.log
grep -E "passed|failed" $T/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -1 $T/smoke.log
timeout 900 python bench.py > $T/bench_default.out 2> $T/bench_default.err; tail -1 $T/bench_default.out > $T/bench_default_line.json
cp gpurun_out/bench_full.json $T/bench_default_full.json 2>/dev/null
python tools/show_bench.py $T/bench_default_line.json 2>/dev/null | head -40
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29661 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 600 python bench.py --force-sharded --chunks 4 --steps 100 --warmup 30 2> $T/fs.err | tail -1 > $T/bench_force_sharded_line.json
cp gpurun_out/bench_full.json $T/bench_force_sharded_full.json 2>/dev/null
timeout 600 python bench.py --config5-share --scaling weak --steps 10 --warmup 3 2> $T/c5.err | tail -1 > $T/bench_config5_share_line.json
cp gpurun_out/bench_full.json $T/bench_config5_share_full.json 2>/dev/null
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
head -c 900 $T/bench_config5_share_line.json; echo
bash tools/gpu_profiles.sh > $T/profiles.log 2>&1; tail -14 $T/profiles.log
exit 0
