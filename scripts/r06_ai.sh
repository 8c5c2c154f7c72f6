#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ai; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_coverage.py -x -q 2>&1 | tail -4 | tee $o/tests.txt
python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "cin8|conv total" | tee $o/cin8.txt
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_plainst.so" 2>&1 | grep "^batch" | tee $o/ab.txt
