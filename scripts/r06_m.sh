#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06m; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; grep -E "passed|failed|Error" $o/pytest.log | tail -5
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r6a.so" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | grep "^batch" | tee $o/ab.txt
