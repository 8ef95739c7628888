#!/usr/bin/env bash
set -u
out=gpurun_out/r4b6; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_order_free.py -x -q 2>&1 | tail -5 | tee $out/pytest.txt
for v in -1 65; do
  MGCFD_EXP_VARIANT=$v timeout -k 10 300 python3 tools/exp/time_sweep.py 2000 25 fast 2>&1 | grep "sweep\|vcycle" | tee -a $out/times.txt
  MGCFD_FREE_NO_ROLES=1 MGCFD_EXP_VARIANT=$v timeout -k 10 300 python3 tools/exp/time_sweep.py 2000 25 fast 2>&1 | grep "sweep\|vcycle" | sed 's/^/noroles /' | tee -a $out/times.txt
done
MGCFD_PLAN_TIMING=1 python3 tools/driver_at_scale.py > $out/driver.txt 2>&1; grep "process wall\|read in\|mgcfd create\|300763" $out/driver.txt | cut -c1-160 | head -40
