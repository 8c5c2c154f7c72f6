#!/bin/bash
# round 5: the unrolled patch kernel with 4x4 taps (layer_4 backward-data passes): parity, layer times, step A/B against the knob off
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/patch4; mkdir -p $o
timeout 900 python -m pytest -x -q --timeout 600 tests/test_gpu_ops.py > $o/tests.log 2>&1; tail -2 $o/tests.log
timeout 900 python -m pytest -x -q --timeout 600 tests/test_gpu_step.py tests/test_gpu_coverage.py tests/test_gpu_soak.py tests/test_gpu_fullwidth.py > $o/tests2.log 2>&1; tail -2 $o/tests2.log
timeout 300 python scripts/layer_profile.py 2>/dev/null | grep -E "conv total|layer_4" > $o/layers.txt; cat $o/layers.txt
bash scripts/ab.sh -b "32 8" "" "tune:patch4=0" 2>&1 | grep "^batch" | tee $o/ab.txt
