#!/bin/bash
T=gpurun_out/r6a
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -s > $T/pytest6.log 2>&1; echo "rc=$?" >> $T/pytest6.log
tail -15 $T/pytest6.log
timeout 300 python tools/host_route_probe.py > $T/host_route.txt 2>&1
cat $T/host_route.txt
exit 0
