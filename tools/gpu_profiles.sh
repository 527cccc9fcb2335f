#!/bin/bash
# round-3 profile set: rocprofv3 kernel traces + PMC passes (separate passes, never combined with traces), summarised ON THE
# BOX by tools/profile_run.py (the csrc fingerprint in every summary is the one of the sources the measurement ran on)
P=gpurun_out/profiles
mkdir -p $P
run() { name=$1; shift; timeout 900 python tools/profile_run.py $name "$@" > gpurun_out/prof_$name.log 2>&1; echo "$name rc=$?"; }
run r03_config2_sat_major --pmc --
run r03_config2_time_major --pmc -- --layout time
run r03_config3_sat_major --pmc -- --deep 1522
run r03_config3_time_major --pmc -- --deep 1522 --layout time
run r03_config2_ecef_sat_major -- --mode ecef
run r03_config2_ecef_time_major -- --layout time --mode ecef
run r03_config5_share --pmc --steps 10 -- --config5-share
run r03_config5_share_f32arith --pmc --steps 10 -- --config5-share --f32-arith
run r03_config5_share_fp64 --pmc --steps 10 -- --config5-share --f32-fp64
cp $P/r03_config2_sat_major.json $P/latest_pmc.json
ls -la $P | tail -20
