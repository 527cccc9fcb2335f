#!/bin/bash
# round-6 profile set: rocprofv3 kernel traces + PMC passes (separate passes, never combined with traces), summarised ON THE
# BOX by tools/profile_run.py (the csrc fingerprint in every summary is the one of the sources the measurement ran on)
P=gpurun_out/profiles
mkdir -p $P
run() { name=$1; shift; timeout 900 python tools/profile_run.py $name "$@" > gpurun_out/prof_$name.log 2>&1; echo "$name rc=$?"; }
run r06_config2_sat_major --pmc --
run r06_config2_time_major --pmc -- --layout time
run r06_config2_time_major_jdfr -- --layout time --grid jdfr
run r06_config2_sat_major_jdfr -- --grid jdfr
run r06_config2_time_major_random --pmc -- --layout time --grid random
run r06_config2_sat_major_random -- --grid random
run r06_config3_sat_major --pmc -- --deep 1522
run r06_config3_time_major --pmc -- --deep 1522 --layout time
run r06_config5_share --pmc --steps 10 -- --config5-share
run r06_fused_screen --pmc --script tools/screen_probe.py --
run r06_screen_all --pmc --script tools/screen_all_probe.py --
cp $P/r06_config2_sat_major.json $P/latest_pmc.json
ls -la $P | tail -30
rm -rf gpurun_out/prof_raw
