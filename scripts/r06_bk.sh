#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bk; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_dp.py -x -q 2>&1 | grep -E " passed| failed|Error|FAILED|assert" | tail -5
for b in 32 8 4; do timeout 300 python scripts/exp_dp1.py $b 2>&1 | grep -E "^bs.*single" | tee -a $o/dp1.txt; done
