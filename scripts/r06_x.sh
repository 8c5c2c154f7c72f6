#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06x; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "tune:wgrad_slab_x100=150" "tune:wgrad_slab_x100=100" "tune:wgrad_slab_x100=200" "tune:wgrad_slab_tile_x1000=20" "tune:wgrad_slab_tile_x1000=50" "tune:wgrad_slab_tile_x1000=100" "tune:wgrad_slab_tile_x1000=150" "tune:wgrad_slab_x100=100 tune:wgrad_slab_tile_x1000=30" 2>&1 | grep "^batch" | tee $o/ab.txt
