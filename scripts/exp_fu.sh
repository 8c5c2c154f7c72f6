#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/fu; mkdir -p $o
for i in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export VP_NO_FUSED_UPDATE=1; else unset VP_NO_FUSED_UPDATE; fi
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile > $o/b.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b.json'));print('no_fused=$v',d['ms_per_step'])"
done; done
