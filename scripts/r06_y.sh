#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06y; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "tune:wgrad_slab_tile_x1000=150" "tune:wgrad_slab_tile_x1000=200" "tune:wgrad_slab_tile_x1000=300" "tune:wgrad_slab_tile_x1000=450" "tune:wgrad_slab_tile_x1000=200 tune:wgrad_fixed_x10=40" "tune:wgrad_slab_tile_x1000=200 tune:wgrad_fixed_x10=160" 2>&1 | grep "^batch" | tee $o/ab.txt
