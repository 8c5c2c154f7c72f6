#!/bin/bash
# round 5: 256 x 256 tile of wgrad_tr.hip: tests, weight-gradient layer times with the knob on / off, step A/B
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/wg; mkdir -p $o
timeout 900 python -m pytest -x -q --timeout 600 tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_fullwidth.py > $o/tests.log 2>&1; grep -E "passed|failed" $o/tests.log
timeout 300 python scripts/layer_profile.py 2>/dev/null | grep -E "conv total|wgrad" | head -24 > $o/layers_on.txt; cat $o/layers_on.txt
bash scripts/ab.sh -b "32 8" "" "tune:wgrad_big=0" 2>&1 | grep "^batch" | tee $o/ab.txt
