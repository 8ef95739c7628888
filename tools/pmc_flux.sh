#!/usr/bin/env bash
# Collect PMC counters for the flux kernel, one rocprofv3 pass per counter group
# (no trace domains combined with --pmc other than --kernel-trace).  Usage: tools/pmc_flux.sh <tag> [flux_only args]
set -u
export TMPDIR=/tmp
tag=$1; shift
groups=(
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY"
 "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_REQ_sum"
)
i=0
for c in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 tools/flux_only.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "failed: $c"
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* > gpurun_out/pmc_${tag}_summary.txt 2>&1
cat gpurun_out/pmc_${tag}_summary.txt
