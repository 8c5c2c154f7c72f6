#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/sk; mkdir -p $o
for cfg in "512 32 2" "256 16 4" "128 8 4" "128 16 8" "64 4 8" "256 8 8"; do
  set -- $cfg
  for gb in 4 32; do
    VP_SPLITK_TARGET=$1 VP_SPLITK_MAX=$2 VP_SPLITK_MINCHUNK=$3 timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --global-batch $gb > $o/b.json 2> $o/b.err
    python -c "
import json;d=json.load(open('$o/b.json'));print('target $1 max $2 minchunk $3 bs$gb',d['ms_per_step'])"
  done
done
