#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bb; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -k "one_output_channel" -s 2>&1 | grep -E " passed| failed|Error|FAILED|worst|assert" | tail -8
for b in 32 8 4; do for s in "" "tune:cout1_wgrad_rows=256" "tune:cout1_wgrad_rows=1024" "tune:cout1_bwd=0"; do
  echo "== batch $b layer_5 wgrad [$s]" | tee -a $o/layers2.txt
  python scripts/layer_profile.py $b 256 bf16 $s 2>/dev/null | grep -E "layer_5:wgrad" | tee -a $o/layers2.txt
done; done
