#!/bin/bash
# round 6, second block: the whole GPU tier, the default bench line, the N > 1-shaped lines on one GPU, the fp32 waves/SIMD A/B,
# the screens' profiles
T=gpurun_out/r6b
mkdir -p $T gpurun_out/profiles
timeout 1500 python -m pytest tests -m gpu -q -x > $T/pytest.log 2>&1; echo "rc=$?" >> $T/pytest.log
tail -4 $T/pytest.log
timeout 900 python bench.py > $T/bench_default.out 2> $T/bench_default.err; tail -1 $T/bench_default.out > $T/bench_default_line.json
cp gpurun_out/bench_full.json $T/bench_default_full.json 2>/dev/null
python tools/show_bench.py $T/bench_default_line.json 2>/dev/null | head -60
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29661 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 600 python bench.py --force-sharded --chunks 4 --steps 100 --warmup 30 2> $T/fs.err | tail -1 > $T/bench_force_sharded_line.json
cp gpurun_out/bench_full.json $T/bench_force_sharded_full.json 2>/dev/null
timeout 600 python bench.py --config5-share --scaling weak --steps 10 --warmup 3 2> $T/c5.err | tail -1 > $T/bench_config5_share_line.json
cp gpurun_out/bench_full.json $T/bench_config5_share_full.json 2>/dev/null
head -c 1500 $T/bench_force_sharded_line.json; echo; head -c 1500 $T/bench_config5_share_line.json; echo
# fp32 share: waves/SIMD of the mixed kernel (same box)
for rep in 1 2; do python tools/sweep.py run --config5-share --steps 10 --warmup 3; done 2>&1 | tee $T/f32_waves.txt
# the screens: kernel trace + counters
timeout 900 python tools/profile_run.py r06_fused_screen --pmc --script tools/screen_probe.py -- > $T/prof_fused.log 2>&1; tail -5 $T/prof_fused.log
timeout 900 python tools/profile_run.py r06_screen_all --pmc --script tools/screen_all_probe.py -- > $T/prof_all.log 2>&1; tail -5 $T/prof_all.log
rm -rf gpurun_out/prof_raw
exit 0
