#!/bin/bash
# experiment: unrolled 3x3 patch kernel tile choices
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/exp_patch3
mkdir -p $o
for p in 3 2 1 0; do
  VP_PATCH2_SMALL=$p timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_small_$p.txt 2>&1
  echo "small=$p"; grep "conv total" $o/layers_small_$p.txt; grep "patch" $o/layers_small_$p.txt | head -16
done
