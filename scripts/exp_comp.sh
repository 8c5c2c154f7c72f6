#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py tests/test_gpu_single_ops.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
bash scripts/kstats.sh | grep -E "composite"
timeout 600 python scripts/soak2.py 20 8 2>&1 | tail -1
