#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06be; mkdir -p $o
for i in 1 2; do python scripts/exp_engine_sequence.py 2>&1 | grep -v amdgpu.ids | tee -a $o/seq.txt; done
python scripts/exp_engine_sequence.py keep 2>&1 | grep -v amdgpu.ids | tee -a $o/seq.txt
