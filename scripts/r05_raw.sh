#!/bin/bash
# round 5: first layers' raw outputs not stored: tests + step A/B against the previous build
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/raw; mkdir -p $o
timeout 1200 python -m pytest -x -q --timeout 600 tests/test_gpu_step.py tests/test_gpu_fullwidth.py tests/test_gpu_soak.py > $o/tests.log 2>&1; tail -5 $o/tests.log
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$GRAFT_REPO_ROOT/voicepuppet_amd/libvp_head.so" 2>&1 | grep "^batch" | tee $o/ab.txt
