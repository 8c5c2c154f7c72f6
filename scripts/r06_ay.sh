#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ay; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -k "one_output_channel or backward_sums" -s 2>&1 | grep -E " passed| failed|Error|FAILED|worst|assert" | tail -8
for b in 32 8; do for s in "" "tune:cout1_bwd=0"; do
  echo "== batch $b layer_5 passes [$s]" | tee -a $o/layers.txt
  python scripts/layer_profile.py $b 256 bf16 $s 2>/dev/null | grep -E "layer_5|conv total" | tee -a $o/layers.txt
done; done
bash scripts/ab.sh -b "32 8 4" "" "tune:cout1_bwd=0" 2>&1 | tee $o/ab.txt
