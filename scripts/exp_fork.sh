#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/exp_fork
mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_step.py -x -q -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed" $o/pytest.log
for p in "d_beside_vgg=0" "d_beside_vgg=1" "d_backward_fork=1" "d_backward_fork=0"; do
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-profile --tune $p > $o/bench_$p.json 2> $o/bench_$p.err
  python -c "
import json;d=json.load(open('$o/bench_$p.json'));print('$p',d['ms_per_step'])"
done
