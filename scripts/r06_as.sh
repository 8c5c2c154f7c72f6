#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06as; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_ops.py -x -q 2>&1 | grep -E " passed| failed|Error|FAILED" | tail -3
bash scripts/ab.sh -b "32 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_prev.so" 2>&1 | tee $o/ab.txt
