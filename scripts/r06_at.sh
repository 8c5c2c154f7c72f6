#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06at; mkdir -p $o
python scripts/exp_host_bound.py 4 8 32 2>&1 | grep -v amdgpu.ids | tee $o/host_bound.txt
