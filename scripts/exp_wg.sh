#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/wg; mkdir -p $o
for cfg in "8 0.15" "16 0.3" "32 0.5" "64 1.0" "16 1.0"; do
  set -- $cfg
  VP_WG_FIXED=$1 VP_WG_SLAB=$2 timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers.txt 2>&1
  echo "fixed=$1 slab=$2 $(grep 'conv total' $o/layers.txt) wgrad_sum=$(grep wgrad $o/layers.txt | awk '{s+=$3} END {print s}')"
  VP_WG_FIXED=$1 VP_WG_SLAB=$2 timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile > $o/b.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b.json'));print('   step',d['ms_per_step'])"
done
