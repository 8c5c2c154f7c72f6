#!/bin/bash
# rocprofv3 kernel stats of the bench step (no CPU baseline, no f32, no event profiling), single stream (--tune streams=1) unless $1 = on
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/kstats
rm -rf $o; mkdir -p $o
one="--tune streams=1"; if [ "$1" = "on" ]; then one=""; fi
rocprofv3 --kernel-trace --stats -d $o/r -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling $one > $o/r.log 2>&1
rm -f $o/r/*kernel_trace.csv
python3 scripts/kstats_summary.py $o/r/r_kernel_stats.csv 13
