#!/bin/bash
# experiment: unrolled 3x3 patch kernel (conv_patch3.hip) vs the generic one
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/exp_patch3
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_fwd or conv_bwd_data or patch" > $o/pytest_ops.log 2>&1
tail -3 $o/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py -x -q -m gpu > $o/pytest_step.log 2>&1
tail -3 $o/pytest_step.log
for p in 1 0; do
  if [ $p = 0 ]; then export VP_NO_PATCH3=1; fi
  timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_p3_$p.txt 2>&1
  grep "conv total" $o/layers_p3_$p.txt; grep "patch" $o/layers_p3_$p.txt | head -16
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-profile > $o/bench_$p.json 2> $o/bench_$p.err
  python -c "
import json;d=json.load(open('$o/bench_$p.json'));print('patch3=$p',d['ms_per_step'])"
done
