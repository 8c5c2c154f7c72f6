#!/bin/bash
# usage: bash scripts/pmc_passes.sh <tag> -- runs scripts/conv_pmc.py under several counter sets
tag=${1:-pmc}
export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TA_TA_BUSY_sum" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_p$i -o p --output-format csv -- python3 scripts/conv_pmc.py 4 > gpurun_out/${tag}_p$i.log 2>&1
  python3 scripts/pmc_table.py gpurun_out/${tag}_p$i igemm_dma > gpurun_out/${tag}_p$i.txt 2>&1
  rm -rf gpurun_out/${tag}_p$i
done
cat gpurun_out/${tag}_p*.txt > gpurun_out/${tag}_all.txt
