#!/bin/bash
# round-5 profile set: rocprofv3 kernel traces + PMC passes (separate passes, never combined with traces), summarised ON THE
# BOX by tools/profile_run.py (the csrc fingerprint in every summary is the one of the sources the measurement ran on)
P=gpurun_out/profiles
mkdir -p $P
run() { name=$1; shift; timeout 900 python tools/profile_run.py $name "$@" > gpurun_out/prof_$name.log 2>&1; echo "$name rc=$?"; }
run r05_config2_sat_major --pmc --
run r05_config2_time_major --pmc -- --layout time
run r05_config2_time_major_jdfr --pmc -- --layout time --grid jdfr
run r05_config2_sat_major_jdfr -- --grid jdfr
run r05_config2_time_major_jitter -- --layout time --grid jitter
run r05_config2_sat_major_jitter -- --grid jitter
run r05_config2_time_major_random -- --layout time --grid random
run r05_config2_sat_major_random -- --grid random
run r05_config3_sat_major --pmc -- --deep 1522
run r05_config3_time_major --pmc -- --deep 1522 --layout time
run r05_config3_time_major_jdfr -- --deep 1522 --layout time --grid jdfr
run r05_config5_share --pmc --steps 10 -- --config5-share
ASTROZ_AMD_COLS=1 run r05_config2_time_major_cols --pmc -- --layout time
ASTROZ_AMD_COLS=1 run r05_config3_time_major_cols -- --deep 1522 --layout time
cp $P/r05_config2_sat_major.json $P/latest_pmc.json
ls -la $P | tail -30
rm -rf gpurun_out/prof_raw
