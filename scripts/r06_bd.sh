#!/bin/bash
# how often does a process get a slow few-frame step, and is it the HIP runtime's hardware-queue assignment (GPU_MAX_HW_QUEUES, default 4)?
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bd; mkdir -p $o
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  for q in 4 8; do
    r=$(GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-profile --no-f32 --no-scaling-ceiling --no-input-pipeline --no-bfmnet-train --no-secondary --global-batch 4 2>/dev/null | python -c "import json,sys; print(round(json.load(sys.stdin)['ms_per_step'],3))")
    echo "run $i GPU_MAX_HW_QUEUES=$q batch 4: $r ms" | tee -a $o/queues.txt
  done
done
