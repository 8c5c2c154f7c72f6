#!/bin/bash
# launch census + kernel stats of the overlapped step at the given global batches: census.sh "4 32"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for gb in ${1:-4 32}; do
  o=gpurun_out/census_bs$gb
  rm -rf $o; mkdir -p $o
  rocprofv3 --kernel-trace --stats -d $o/r -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --global-batch $gb > $o/r.log 2>&1
  rm -f $o/r/*kernel_trace.csv
  echo "== global batch $gb"; python3 scripts/kstats_summary.py $o/r/r_kernel_stats.csv 13 | head -${2:-45}
done
