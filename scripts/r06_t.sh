#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06t; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:igemm_splitk_target=64" "tune:igemm_splitk_target=64 tune:patch_min_blocks=256" "tune:patch_min_blocks=256" "tune:igemm_splitk_target=64 tune:patch_min_blocks=192" 2>&1 | grep "^batch" | tee $o/ab.txt
