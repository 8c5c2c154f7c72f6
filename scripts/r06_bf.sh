#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bf; mkdir -p $o
for q in 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q, earlier engines kept alive" | tee -a $o/seq_queues.txt
  GPU_MAX_HW_QUEUES=$q python scripts/exp_engine_sequence.py keep 2>&1 | grep -v amdgpu.ids | tee -a $o/seq_queues.txt
done
