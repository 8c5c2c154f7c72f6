#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bh; mkdir -p $o
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E " passed| failed|Error|FAILED|assert" | tail -5
python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "layer_5|conv total" | tee $o/layers.txt
bash scripts/ab.sh -b "32 8 4" "" "tune:cout1_bwd=0" 2>&1 | tee $o/ab.txt
