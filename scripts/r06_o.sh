#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06o; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:igemm_small_grid=32" "tune:igemm_small_grid=64" "tune:igemm_small_grid=128" "tune:igemm_small_grid=255" 2>&1 | grep "^batch" | tee $o/ab.txt
