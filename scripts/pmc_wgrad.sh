#!/bin/bash
# SQ counter passes on isolated weight-gradient launches
tag=${1:-pmcwg}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python3 scripts/wgrad_pmc.py 10 > gpurun_out/${tag}_plain.txt 2>&1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_p$i -o p --output-format csv -- python3 scripts/wgrad_pmc.py 3 > gpurun_out/${tag}_p$i.log 2>&1
  python3 scripts/pmc_table.py gpurun_out/${tag}_p$i wgrad_tr > gpurun_out/${tag}_p$i.txt 2>&1
  rm -rf gpurun_out/${tag}_p$i
done
cat gpurun_out/${tag}_plain.txt gpurun_out/${tag}_p*.txt > gpurun_out/${tag}_all.txt
