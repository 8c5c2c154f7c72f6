#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/bs; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_ops.py -x -q -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed" $o/pytest.log
for gb in 32 16 8 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --global-batch $gb > $o/b$gb.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b$gb.json'));print('bs$gb',d['ms_per_step'], d['value'])"
done
