#!/usr/bin/env bash
set -u
E=mg-cfd-app-plain_amd/csrc/build/exp
out=gpurun_out/r4b4; mkdir -p $out
for v in base v2 v2_o1sc1 v2_o2sc1 v2_o3sc1 o2sc1; do
  MGCFD_LIB=$E/libmgcfd_hip_$v.so timeout -k 10 120 python3 tools/exp/time_flux.py 67 500 free 2>&1 | grep "median" | tee -a $out/times.txt
done
MGCFD_LIB=$E/libmgcfd_hip_v2_o2sc1.so timeout -k 10 120 python3 tools/exp/time_flux.py 96 300 free 2>&1 | grep "median" | tee -a $out/times.txt
MGCFD_LIB=$E/libmgcfd_hip_base.so timeout -k 10 120 python3 tools/exp/time_flux.py 96 300 free 2>&1 | grep "median" | tee -a $out/times.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_order_free.py -x -q 2>&1 | tail -5 | tee $out/pytest.txt
