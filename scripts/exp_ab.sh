#!/bin/bash
# A/B of one vp_tune knob on the full step: bash scripts/exp_ab.sh key v0 v1 [repeats]
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/ab; mkdir -p $o
for i in $(seq 1 ${4:-3}); do
for v in $2 $3; do
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --tune $1=$v > $o/b.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b.json'));print('$1=$v',d['ms_per_step'])"
done; done
