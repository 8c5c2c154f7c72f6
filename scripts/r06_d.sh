#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06d; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -k "epilogue_in_situ" -s 2>&1 | grep -v amdgpu.ids | tail -15 | tee $o/t1.txt
timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_fullwidth.py tests/test_gpu_coverage.py -x -q 2>&1 | tail -8 | tee $o/t2.txt
bash scripts/ab.sh -b "32 8 4" "" "tune:bwd_sums_in_epilogue=0" "tune:bwd_sums_in_epilogue=0 tune:vgg_real_fork=0" 2>&1 | grep "^batch" | tee $o/ab.txt
