#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06w; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:wgrad_slab_x100=150" "tune:wgrad_slab_x100=300" "tune:wgrad_slab_x100=600" "tune:wgrad_slab_x100=1200" "tune:wgrad_slab_x100=300 tune:wgrad_fixed_x10=40" 2>&1 | grep "^batch" | tee $o/ab.txt
