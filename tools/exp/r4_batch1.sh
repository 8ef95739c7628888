#!/usr/bin/env bash
# round 4, batch 1: cache-policy and load-order variants of the standalone flux launch (tools/exp_flags.py builds)
set -u
export TMPDIR=/tmp
E=mg-cfd-app-plain_amd/csrc/build/exp
out=gpurun_out/r4b1; mkdir -p $out
for v in base stnt stsc1 ldnt ldnt_stnt ldnt_stsc1 ord0 ord1 ord2 ord3 base; do
  MGCFD_LIB=$E/libmgcfd_hip_$v.so timeout -k 10 120 python3 tools/exp/time_flux.py 67 500 exact,free 2>&1 | grep -v "^half rows" | tee -a $out/times.txt
done
for v in base stsc1 ldnt ldnt_stsc1; do
  export MGCFD_LIB=$E/libmgcfd_hip_$v.so
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_EA0_WRREQ_sum"; do
    timeout -k 10 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${v}_$i -- python3 tools/flux_only.py --fast --variant 65 --launches 50 > $out/pmc_${v}_$i.log 2>&1 || echo "pmc failed: $v $c"
    i=$((i+1))
  done
  python3 tools/pmc_summary.py $out/pmc_${v}_[0-9] > $out/pmc_${v}_summary.txt 2>&1
  echo "== $v"; grep "k_flux_free" $out/pmc_${v}_summary.txt
done
