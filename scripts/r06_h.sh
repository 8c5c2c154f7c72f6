#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06h; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_fullwidth.py tests/test_gpu_coverage.py tests/test_gpu_training_trajectory.py -x -q -s 2>&1 | grep -E "passed|failed|Error|error|update norms|band|bwd sums" | tail -60 | tee $o/tests.txt
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r6a.so" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | grep "^batch" | tee $o/ab.txt
