#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ad; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "tune:igemm_split_model=0" "" "tune:igemm_split_L=32" "tune:igemm_split_L=8" "tune:igemm_split_R=256" "tune:igemm_split_cap=8" "tune:igemm_small_grid=0" "tune:igemm_split_D_kb=500" 2>&1 | grep "^batch" | tee $o/ab.txt
