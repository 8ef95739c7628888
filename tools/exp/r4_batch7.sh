#!/usr/bin/env bash
set -u
out=gpurun_out/r4b7; mkdir -p $out
timeout -k 10 300 python3 tools/exp/time_sweep.py 200 25 exact,fast 2>&1 | grep "vcycle" | tee -a $out/times.txt
MGCFD_SETS_GROW=1 MGCFD_NO_BOXES=1 timeout -k 10 300 python3 tools/exp/time_sweep.py 200 25 exact,fast 2>&1 | grep "vcycle" | sed 's/^/sets-grown /' | tee -a $out/times.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_partitioned_cycles.py -x -q 2>&1 | tail -3
