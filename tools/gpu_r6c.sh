#!/bin/bash
# round 6: the screens after the launch fold / zero-copy results / XCD-local cell kernels / persistent scratch
T=gpurun_out/r6c
mkdir -p $T gpurun_out/profiles
timeout 900 python -m pytest tests -m gpu -q -x -k "screen or coarse or highlevel or c_client or group" > $T/pytest_screen.log 2>&1; echo "rc=$?" >> $T/pytest_screen.log
tail -3 $T/pytest_screen.log
python tools/screen_probe.py | tee $T/screen_probe.txt
python tools/screen_all_probe.py | tee $T/screen_all_probe.txt
timeout 900 python tools/profile_run.py r06_fused_screen --pmc --script tools/screen_probe.py -- > $T/prof_fused.log 2>&1; tail -3 $T/prof_fused.log
timeout 900 python tools/profile_run.py r06_screen_all --pmc --script tools/screen_all_probe.py -- > $T/prof_all.log 2>&1; tail -3 $T/prof_all.log
head -24 gpurun_out/profiles/r06_screen_all.txt
rm -rf gpurun_out/prof_raw
exit 0
