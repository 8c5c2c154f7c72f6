#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bo; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_dp.py -x -q 2>&1 | grep -E " passed| failed|Error|FAILED|assert" | tail -5
bash scripts/r06_bn.sh
python scripts/exp_engine_sequence.py keep 2>&1 | grep -v amdgpu.ids | tee $o/seq_keep.txt
