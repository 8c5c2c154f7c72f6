#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06i; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_soak.py tests/test_gpu_fullwidth.py tests/test_gpu_coverage.py tests/test_gpu_training_trajectory.py -x -q -s > $o/tests.log 2>&1
grep -n "Aborted\|Fatal\|passed\|failed" $o/tests.log | head; grep -n -B30 "Fatal Python error" $o/tests.log | head -80
