#!/bin/bash
# step time A/B: env settings given as arguments (quoted, e.g. "VP_SMALLP=0" ""), batches in $BATCHES (default "4 8 32"), two rounds each
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/ab
mkdir -p $o
for round in 1 2; do
  for gb in ${BATCHES:-4 8 32}; do
    i=0
    for e in "$@"; do
      i=$((i+1))
      env $e python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --global-batch $gb > $o/b_${gb}_$i.json 2> $o/b_${gb}_$i.err
      python -c "import json;d=json.load(open('$o/b_${gb}_$i.json'));print('batch $gb [$e]', round(d['ms_per_step'],3))" || tail -3 $o/b_${gb}_$i.err
    done
  done
done
