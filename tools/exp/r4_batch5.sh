#!/usr/bin/env bash
set -u
E=mg-cfd-app-plain_amd/csrc/build/exp
out=gpurun_out/r4b5; mkdir -p $out
for v in cur stg_sc1 stg_nt cur; do
  MGCFD_LIB=$E/libmgcfd_hip_$v.so timeout -k 10 200 python3 tools/exp/time_sweep.py 2000 25 2>&1 | grep "sweep\|vcycle" | tee -a $out/times.txt
done
MGCFD_LIB=$E/libmgcfd_hip_cur.so timeout -k 10 120 python3 tools/exp/time_flux.py 67 500 exact,free 2>&1 | grep "median" | tee -a $out/times.txt
