#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ba; mkdir -p $o
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E " passed| failed|Error|FAILED|assert" | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-80
bash scripts/ab.sh -b "32 8 4" "" "tune:cout1_bwd=0" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | tee $o/ab.txt
