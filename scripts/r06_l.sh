#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06l; mkdir -p $o
timeout 1500 python scripts/train_spread.py 200 $o/train_spread.json 2>&1 | grep -v amdgpu.ids | tee $o/train_spread.txt
