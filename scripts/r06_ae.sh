#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ae; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:igemm_long_k_chunks=64" "tune:igemm_long_k_chunks=128" "tune:igemm_long_k_chunks=256" 2>&1 | grep "^batch" | tee $o/ab.txt
