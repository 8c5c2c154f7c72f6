#!/bin/bash
# kernel timeline of the overlapped step: trace kept (gzip) for scripts/timeline.py;  $1 = global batch (default 32)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
gb=${1:-32}
o=gpurun_out/timeline
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o/on -o on --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary --global-batch $gb > $o/on.log 2>&1
find $o -name "*kernel_trace.csv" -exec gzip -9 {} \;
tail -c 300 $o/on.log
