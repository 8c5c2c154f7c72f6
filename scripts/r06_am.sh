#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06am; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "thin or conv_bwd_data" 2>&1 | grep -E " passed| failed|Error" | tail -3
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py -x -q 2>&1 | grep -E " passed| failed|Error" | tail -3
python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "cout8|cout4|dcout8|conv total" | tee $o/thin.txt
