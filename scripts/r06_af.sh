#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06af; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:igemm_small_grid_kmax=4096" "tune:igemm_small_grid_kmax=2048" 2>&1 | grep "^batch" | tee $o/ab.txt
