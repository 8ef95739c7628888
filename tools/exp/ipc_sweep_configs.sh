#!/usr/bin/env bash
# A sequence of DIFFERENT rank counts / level sizes / stage forms through tools/ipc_ranks_check.py (one GPU, the ranks sharing
# it; one configuration at a time, stops at the first failure).  Usage (GPU box): bash tools/exp/ipc_sweep_configs.sh <log>
log=${1:-gpurun_out/ipc_sweep.log}
: > "$log"
n=0
for mesh in m6wing fvcorr tet; do
  for ranks in 2 3 4 5; do
    for form in "" "--unsplit" "--fused" "--fused --one-by-one"; do
      case $mesh in tet) lat=$((20 + 7 * ranks));; *) lat=$((12 + 3 * ranks + n % 5));; esac
      sweeps=$((4 + n % 6))
      n=$((n + 1))
      echo "== $mesh ranks=$ranks lattice=$lat sweeps=$sweeps $form" >> "$log"
      timeout -k 10 120 python tools/ipc_ranks_check.py --mesh $mesh --ranks $ranks --lattice $lat --sweeps $sweeps $form >> "$log" 2>&1 || { echo "FAILED at configuration $n" | tee -a "$log"; exit 1; }
    done
  done
done
echo "configurations passed: $n" | tee -a "$log"
