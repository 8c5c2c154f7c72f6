#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/co8; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py tests/test_gpu_ops.py -x -q -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed|Error" $o/pytest.log | tail -3
timeout 300 python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "conv1_1|layer_1|conv total"
VP_NO_DCOUT8=1 VP_NO_COUT8=1 timeout 300 python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "conv1_1:bwd|layer_1:bwdG|conv total"
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
