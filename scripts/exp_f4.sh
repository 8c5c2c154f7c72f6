#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/f4; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_bfmnet_train.py -x -q -m gpu -s > $o/pytest.log 2>&1; tail -40 $o/pytest.log
