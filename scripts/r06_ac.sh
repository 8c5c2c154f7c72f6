#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ac; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:wgrad_big=0" "tune:wgrad_fixed_x10=40" "tune:wgrad_fixed_x10=160" "tune:wgrad_slab_x100=0" "tune:wgrad_slab_x100=60" 2>&1 | grep "^batch" | tee $o/ab.txt
