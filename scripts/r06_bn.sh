#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bn; mkdir -p $o
for b in 4 32; do for v in pg_first reserved_first engine_first reserved_first pg_first; do
  timeout 300 python scripts/exp_dp_order.py $v $b 2>&1 | grep -E "batch" | tee -a $o/order2.txt
done; done
