#!/bin/bash
# quick loop: step tests + layer times of a pattern + step A/B against the previous build (libvp_head.so)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/quick; mkdir -p $o
timeout 900 python -m pytest -x -q --timeout 600 tests/test_gpu_step.py tests/test_gpu_coverage.py tests/test_gpu_soak.py > $o/tests.log 2>&1; tail -2 $o/tests.log
timeout 300 python scripts/layer_profile.py 2>/dev/null | grep -E "conv total|$1" > $o/layers.txt; cat $o/layers.txt
bash scripts/ab.sh -b "32" "" "VP_LIB=$GRAFT_REPO_ROOT/voicepuppet_amd/libvp_head.so" 2>&1 | grep "^batch" | tee $o/ab.txt
