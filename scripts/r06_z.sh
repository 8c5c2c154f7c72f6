#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06z; mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_ops.py -x -q -k "k_split_of_rounds or thin_layer" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; grep -E "passed|failed|Error|^FAILED|AssertionError" $o/pytest.log | tail -12
