#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06aa; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:smallp_split_target=192" "tune:smallp_split_target=768" "tune:igemm_splitk_cap=4" "tune:igemm_splitk_cap=16" "tune:igemm_splitk_minchunk=8" "tune:igemm_splitk_minchunk=2" "tune:wgrad_resident_blocks=256" "tune:wgrad_resident_blocks=1024" 2>&1 | grep "^batch" | tee $o/ab.txt
