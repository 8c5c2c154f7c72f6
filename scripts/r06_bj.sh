#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bj; mkdir -p $o
for b in 32 4; do timeout 300 python scripts/exp_dp1.py $b 2>&1 | grep -E "^bs" | tee -a $o/dp1.txt; done
