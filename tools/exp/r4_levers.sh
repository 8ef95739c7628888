#!/usr/bin/env bash
# round 4: the flux kernel's levers, measured (profiles/r4_flux_levers.txt is this script's output, assembled)
set -u
export TMPDIR=/tmp
E=mg-cfd-app-plain_amd/csrc/build/exp
out=gpurun_out/r4lev; mkdir -p $out
echo "== tile-shaped streaming: register loads against LDS-DMA (tools/stream_ceiling.hip --dma)" | tee $out/levers.txt
$E/stream_ceiling --dma 2>&1 | tee -a $out/levers.txt
echo "== event-timed batches of the standalone flux launch, 67^3 level (tools/exp/time_flux.py)" | tee -a $out/levers.txt
for v in cur plainst ord5 h1 h2 h3 allaos cur; do
  MGCFD_LIB=$E/libmgcfd_hip_$v.so timeout -k 10 120 python3 tools/exp/time_flux.py 67 500 exact,free 2>&1 | grep "median" | tee -a $out/levers.txt
done
echo "== counters of k_flux_free per launch (rocprofv3 --pmc, one pass per group; 51 launches)" | tee -a $out/levers.txt
for v in cur h1 h2 h3 allaos; do
  export MGCFD_LIB=$E/libmgcfd_hip_$v.so
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
    timeout -k 10 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${v}_$i -- python3 tools/flux_only.py --fast --variant 65 --launches 50 > $out/pmc_${v}_$i.log 2>&1 || echo "pmc failed: $v $c"
    i=$((i+1))
  done
  python3 tools/pmc_summary.py $out/pmc_${v}_[0-9] > $out/pmc_${v}_summary.txt 2>&1
  echo "-- $v" | tee -a $out/levers.txt; grep "k_flux_free" $out/pmc_${v}_summary.txt | sed 's/void mgcfd::fast:://' | tee -a $out/levers.txt
done
unset MGCFD_LIB
echo "== phase clocks of k_flux_free, current build (tools/phase_half.py, PH_NS=fast)" | tee -a $out/levers.txt
PH_LIB=$PWD/$E/libmgcfd_hip_ph_cur.so python3 tools/phase_half.py run 65 67 2>&1 | grep -v amdgpu.ids | tee -a $out/levers.txt
rm -rf $out/pmc_*_[0-9]
