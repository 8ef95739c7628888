#!/usr/bin/env bash
# does an XCD's L2 keep a level's read-only plan across back-to-back launches?  FETCH_SIZE of the flux launch on small levels
set -u
export TMPDIR=/tmp
out=gpurun_out/r4b3; mkdir -p $out
for L in 24 34 44; do
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    timeout -k 10 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_L${L}_$i -- python3 tools/flux_only.py --fast --variant 65 --launches 50 --lattice $L > $out/pmc_L${L}_$i.log 2>&1 || echo "pmc failed: $L $c"
    i=$((i+1))
  done
  python3 tools/pmc_summary.py $out/pmc_L${L}_[0-9] > $out/pmc_L${L}_summary.txt 2>&1
  echo "== lattice $L"; grep "k_flux_free" $out/pmc_L${L}_summary.txt
done
