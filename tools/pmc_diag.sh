#!/bin/bash
# Stall diagnosis of the split-fp16 forward's kernels: separate rocprofv3 --pmc passes (8 SQ slots / 4 TCC slots per pass)
# of tools/one_forward.py, per-kernel means.  Usage (GPU box, repo root): tools/pmc_diag.sh <out.txt> [kernel substring]
out=${1:-gpurun_out/pmc_diag.txt}; sub=${2:-conv_ws}
root=$PWD; : > "$root/$out.tmp"; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAVES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pd_$i -o p --output-format csv -- python3 $root/tools/one_forward.py f16x2 > /dev/null 2>&1
  python3 $root/tools/pmc_kernels.py /tmp/pd_$i "$sub" >> $root/$out.tmp 2>&1
done
mv $root/$out.tmp $root/$out; cat $root/$out
