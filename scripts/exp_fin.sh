#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/fin; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py tests/test_gpu_single_ops.py -x -q -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed|Error" $o/pytest.log | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --global-batch 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bs8', d['ms_per_step'])"
bash scripts/kstats.sh on | grep -E "finalize"
