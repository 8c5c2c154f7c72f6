#!/bin/bash
# parity tests that exercise the step executor + op entry points, summary line only; then optional extra command
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/check
mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fullwidth.py tests/test_gpu_single_ops.py tests/test_gpu_dp.py -x -q -m gpu > $o/pytest.log 2>&1
grep -E "passed|failed|error" $o/pytest.log | tail -3
grep -E "^E  |Error" $o/pytest.log | head -10
