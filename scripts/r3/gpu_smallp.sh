#!/bin/bash
# parity of the step + op tests with the few-pixel kernel in place, then batch 4 / 8 / 32 step times
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/smallp
mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fullwidth.py tests/test_gpu_single_ops.py -x -q -m gpu > $o/pytest.log 2>&1
tail -15 $o/pytest.log
for gb in 4 8 32; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --global-batch $gb > $o/bench_$gb.json 2> $o/bench_$gb.err
  python -c "import json;d=json.load(open('$o/bench_$gb.json'));print($gb, round(d['ms_per_step'],3))" || tail -5 $o/bench_$gb.err
done
