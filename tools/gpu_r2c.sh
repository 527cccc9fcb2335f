#!/bin/bash
mkdir -p gpurun_out/r2c
L=gpurun_out/r2c/probe.log
ls /sys/class/drm/ > $L 2>&1; ls /sys/class/drm/card*/device/ | head -80 >> $L 2>&1
rocm-smi --showclocks --showpower --showtemp >> $L 2>&1
python tools/clock_probe.py --seconds 2.0 --tag fast >> $L 2>&1
python tools/clock_probe.py --seconds 2.0 --no-fast-path --tag generic >> $L 2>&1
ASTROZ_AMD_LIB=$PWD/tools/variants/lib_stores.so python tools/clock_probe.py --seconds 1.5 --tag stores >> $L 2>&1
ASTROZ_AMD_LIB=$PWD/tools/variants/lib_arith.so python tools/clock_probe.py --seconds 1.5 --tag arith >> $L 2>&1
python tools/clock_probe.py --seconds 1.5 --pos-only --tag fast_pos >> $L 2>&1
python tools/clock_probe.py --seconds 1.5 --layout time --tag fast_tm >> $L 2>&1
python tools/clock_probe.py --seconds 1.0 --idle-ms 1.0 --tag fast_duty >> $L 2>&1
rocm-smi --showclocks --showpower --showtemp >> $L 2>&1
