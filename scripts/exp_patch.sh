#!/bin/bash
# experiment: stride-1 patch kernel tiles; consecutive steps; input pipeline test
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/exp_patch
mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_input_pipeline.py -q -m gpu -s > $o/pytest_ip.log 2>&1; tail -4 $o/pytest_ip.log
timeout 600 python -m pytest tests/test_gpu_fullwidth.py -q -m gpu -s > $o/pytest_fw.log 2>&1; grep -E "step [0-9]:|full width|config|passed|failed|Error" $o/pytest_fw.log
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_fwd or conv_bwd_data or patch" > $o/pytest_ops.log 2>&1
tail -3 $o/pytest_ops.log
for cfg in "7 3" "6 1"; do
  set -- $cfg
  VP_PATCH2=$1 VP_PATCH2_SMALL=$2 timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_p$1_s$2.txt 2>&1
  grep "conv total" $o/layers_p$1_s$2.txt; grep patch $o/layers_p$1_s$2.txt | head -16
done
