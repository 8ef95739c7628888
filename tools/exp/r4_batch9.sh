#!/usr/bin/env bash
set -u
out=gpurun_out/r4b9; mkdir -p $out
for e in 1 0 1 0; do MGCFD_FREE_SECOND_TILE=$e timeout -k 10 120 python3 tools/exp/time_flux.py 67 500 free 2>&1 | grep median | sed "s/^/second_tile=$e /" | tee -a $out/times.txt; done
for L in 70 74; do for e in 1 0; do MGCFD_FREE_SECOND_TILE=$e timeout -k 10 120 python3 tools/exp/time_flux.py $L 500 free 2>&1 | grep median | sed "s/^/second_tile=$e /" | tee -a $out/times.txt; done; done
timeout -k 10 600 python3 -m pytest tests/test_gpu_order_free.py -x -q 2>&1 | tail -3
