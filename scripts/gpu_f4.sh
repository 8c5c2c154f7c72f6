#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/f4
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_bfmnet_train.py -x -q -m gpu > $o/pytest.log 2>&1
tail -25 $o/pytest.log
for b in 4 32; do timeout 300 python scripts/bench_bfmnet_train.py 30 $b > $o/bench_b$b.json 2> $o/bench_b$b.err; cat $o/bench_b$b.json; tail -3 $o/bench_b$b.err; done
