#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/p2; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $o/pytest_ops.log 2>&1; tail -3 $o/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py -x -q -m gpu > $o/pytest_step.log 2>&1; grep -E "passed|failed|Error|error" $o/pytest_step.log | tail -3
for v in 1 0; do
  if [ $v = 0 ]; then export VP_NO_PATCH2=1; else unset VP_NO_PATCH2; fi
  timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_$v.txt 2>&1
  echo "patch2=$v $(grep 'conv total' $o/layers_$v.txt)"
  grep "layer_2:bwd \|layer_3:bwd \|merged2_decoder_2:fwd\|encoder_2:bwd\|layer_2:bwdG\|encoder_fg_2:bwd\|merged2_decoder_3:fwd\|layer_3:bwdG\|encoder_3:bwd\|merged2_decoder_4:fwd" $o/layers_$v.txt
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile > $o/b.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b.json'));print('   step',d['ms_per_step'])"
done
