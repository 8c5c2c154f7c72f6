#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/exp_fork
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py -x -q -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed" $o/pytest.log
for p in "$@"; do
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --tune $p > $o/bench_$p.json 2> $o/bench_$p.err
  python -c "
import json;d=json.load(open('$o/bench_$p.json'));print('$p',d['ms_per_step'])"
done
