#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06aw; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -x -q 2>&1 | grep -E " passed| failed|Error|FAILED" | tail -3
for b in 32 4; do python scripts/layer_profile.py $b 256 bf16 2>/dev/null | grep -E "cin8|conv total" | tee -a $o/cin8.txt; done
bash scripts/ab.sh -b "32 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | tee $o/ab.txt
