#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/xcd; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or bwd_weight or conv_fwd or patch" > $o/pytest_ops.log 2>&1; tail -1 $o/pytest_ops.log
for v in 0 1; do
  if [ $v = 1 ]; then export VP_NO_XCD_REMAP=1; else unset VP_NO_XCD_REMAP; fi
  timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_$v.txt 2>&1
  echo "no_xcd=$v"; grep "conv total" $o/layers_$v.txt; grep "wgrad\|patch" $o/layers_$v.txt | head -26
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-profile > $o/b.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b.json'));print('no_xcd=$v',d['ms_per_step'])"
done
