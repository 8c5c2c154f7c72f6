#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/quick
mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py -x -q -m gpu > $o/pytest.log 2>&1
grep -E "passed|failed|Error" $o/pytest.log | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling > $o/bench.json 2> $o/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/quick/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['frac'])
for k in d['kernels'][:10]: print("%-28s calls %5.1f ms %6.3f TF %7.1f"%(k['name'],k['calls_per_step'],k['ms_per_step'],k['tflops']))
P
