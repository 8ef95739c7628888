#!/usr/bin/env bash
set -u
E=mg-cfd-app-plain_amd/csrc/build/exp
out=gpurun_out/r4b2; mkdir -p $out
for v in base o2sc1 h1 h2 h3 h4 ownaos staos allaos base; do
  MGCFD_LIB=$E/libmgcfd_hip_$v.so timeout -k 10 120 python3 tools/exp/time_flux.py 67 500 free 2>&1 | grep "median" | tee -a $out/times.txt
done
