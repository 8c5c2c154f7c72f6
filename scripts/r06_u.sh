#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06u; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; grep -E "passed|failed|Error|^FAILED" $o/pytest.log | tail -12
timeout 1500 python scripts/train_spread.py 200 $o/train_spread.json 2>&1 | grep -v amdgpu.ids > $o/train_spread.txt; tail -18 $o/train_spread.txt
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | grep "^batch" | tee $o/ab.txt
