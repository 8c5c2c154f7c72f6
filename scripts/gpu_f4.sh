#!/bin/bash
# BFMNet training step on the GPU box: parity tests, then the step time (eager / graph replay, batch 4 / 32; one-stream A/B)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/f4
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_bfmnet_train.py tests/test_gpu_api.py -x -q -m gpu > $o/pytest.log 2>&1
grep -vi warn $o/pytest.log | tail -25
for m in eager graph; do for b in 4 32; do timeout 300 python scripts/bench_bfmnet_train.py 30 $b 35709 $m > $o/bench_${m}_b$b.json 2> $o/bench_${m}_b$b.err; cat $o/bench_${m}_b$b.json; grep -v amdgpu.ids $o/bench_${m}_b$b.err | tail -3; done; done
for b in 4 32; do echo "one stream, batch $b:"; timeout 300 python scripts/bench_bfmnet_train.py 30 $b 35709 graph one 2>/dev/null; done
echo "auto schedule, batch 32:"; timeout 300 python scripts/bench_bfmnet_train.py 30 32 35709 auto 2>/dev/null
