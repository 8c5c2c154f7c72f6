#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06an; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E " passed| failed|Error|FAILED" | tail -5
python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "cout8|cout4|dcout8|cin8|conv total" | tee $o/thin.txt
bash scripts/ab.sh -b "32 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | tail -6 | tee $o/ab32.txt
