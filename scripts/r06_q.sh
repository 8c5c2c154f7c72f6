#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06q; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; grep -E "passed|failed|Error|assert" $o/pytest.log | tail -8
