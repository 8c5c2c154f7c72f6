#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bl; mkdir -p $o
for b in 16 8 4; do for k in "" three; do timeout 300 python scripts/exp_dp1.py $b $k 2>&1 | grep -E "single-GPU" | tee -a $o/dp1_streams.txt; done; done
