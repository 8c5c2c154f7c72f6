#!/bin/bash
# experiment: LDS-DMA + transpose-read weight gradient (wgrad_tr.hip)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/exp_wgrad
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bwd_weight" > $o/pytest_ops.log 2>&1
tail -5 $o/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu > $o/pytest_step.log 2>&1
grep -E "passed|failed" $o/pytest_step.log
for tr in 0 1; do
  VP_WGRAD_TR=$tr timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_tr$tr.txt 2>&1
  grep "conv total" $o/layers_tr$tr.txt; grep wgrad $o/layers_tr$tr.txt | head -12
done
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-profile > $o/bench.json 2> $o/bench.err
python -c "import json;d=json.load(open('$o/bench.json'));print(d['ms_per_step'],d['value'])"
