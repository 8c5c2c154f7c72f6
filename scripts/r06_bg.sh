#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_step.py -x -q 2>&1 | grep -E " passed| failed|Error|FAILED|assert" | tail -5
