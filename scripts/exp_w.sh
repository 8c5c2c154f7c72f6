#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/exp_w
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or bwd_weight" > $o/pytest_ops.log 2>&1
tail -2 $o/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py -x -q -m gpu > $o/pytest_step.log 2>&1
grep -E "passed|failed" $o/pytest_step.log
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers.txt 2>&1
grep "conv total" $o/layers.txt; grep "wgrad" $o/layers.txt | head -30
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile > $o/bench.json 2> $o/bench.err
python -c "
import json;d=json.load(open('$o/bench.json'));print(d['ms_per_step'])"
