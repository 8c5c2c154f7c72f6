#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06p; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; grep -E "passed|failed|Error|assert" $o/pytest.log | tail -8
bash scripts/ab.sh -b "32 8 4" "" "tune:igemm_small_grid=0" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | grep "^batch" | tee $o/ab.txt
