#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/xcd; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or bwd_weight" > $o/pytest_ops.log 2>&1; tail -1 $o/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py -x -q -m gpu > $o/pytest_step.log 2>&1; grep -E "passed|failed" $o/pytest_step.log
for v in 3 1; do
  VP_WGRAD_TR=$v timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_$v.txt 2>&1
  echo "tr=$v"; grep "conv total" $o/layers_$v.txt; grep "wgrad" $o/layers_$v.txt | head -12
  VP_WGRAD_TR=$v timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile > $o/b.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b.json'));print('tr=$v',d['ms_per_step'])"
done
