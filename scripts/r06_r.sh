#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06r; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:igemm_splitk_target=64" "tune:igemm_splitk_target=192" "tune:igemm_splitk_target=256" 2>&1 | grep "^batch" | tee $o/ab.txt
