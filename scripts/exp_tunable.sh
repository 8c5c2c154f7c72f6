#!/bin/bash
# experiment: PyTorch TunableOp (rocBLAS / hipBLASLt solution search per GEMM shape) on the BFMNet training step
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/tunable; mkdir -p $o
for b in 4 32; do
  timeout 300 python scripts/bench_bfmnet_train.py 30 $b 35709 eager 2>/dev/null | cut -c1-140
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 \
  PYTORCH_TUNABLEOP_FILENAME=$o/tunable_b$b.csv PYTORCH_TUNABLEOP_VERBOSE=0 timeout 1200 python scripts/bench_bfmnet_train.py 30 $b 35709 eager 2> $o/err_b$b.log | cut -c1-140
  tail -2 $o/err_b$b.log | cut -c1-200
  wc -l $o/tunable_b$b*.csv
done
