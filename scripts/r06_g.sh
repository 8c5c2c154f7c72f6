#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06g; mkdir -p $o
timeout 1500 python scripts/train_spread.py 200 $o/train_spread.json 2>&1 | grep -v amdgpu.ids | tee $o/train_spread.txt
timeout 900 python -m pytest tests/test_gpu_fullwidth.py -x -q -s -k "batch_of_four" 2>&1 | grep -E "update norms|passed|failed|worst" | tee $o/fullwidth_n4.txt
