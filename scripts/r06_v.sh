#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06v; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:wgrad_slab_x100=50" "tune:wgrad_slab_x100=150" "tune:wgrad_fixed_x10=160" "tune:wgrad_fixed_x10=40" "tune:smallp_max_pixels=512" "tune:smallp_max_pixels=128" 2>&1 | grep "^batch" | tee $o/ab.txt
