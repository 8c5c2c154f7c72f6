#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06s; mkdir -p $o
bash scripts/ab.sh -b "8 4" "" "tune:igemm_splitk_target=64" "tune:igemm_splitk_target=80" "tune:igemm_splitk_target=96" "tune:igemm_splitk_target=112" 2>&1 | grep "^batch" | tee $o/ab.txt
