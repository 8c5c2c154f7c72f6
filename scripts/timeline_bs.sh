#!/bin/bash
# kernel timeline (verbose) of the overlapped step at the given global batches; usage: timeline_bs.sh "4 8"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for gb in ${1:-4 8}; do
  o=gpurun_out/tl_bs$gb
  rm -rf $o; mkdir -p $o
  python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --global-batch $gb > $o/bench.json 2> $o/bench.err
  rocprofv3 --kernel-trace --stats -d $o/on -o on --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --global-batch $gb > $o/on.log 2>&1
  python3 scripts/timeline.py $(find $o -name "*kernel_trace.csv") 6 -v > $o/timeline_v.txt
  python3 scripts/timeline.py $(find $o -name "*kernel_trace.csv") 6 > $o/timeline.txt
  find $o -name "*kernel_trace.csv" -exec gzip -9 {} \;
  python3 -c "import json;d=json.load(open('$o/bench.json'));print($gb, d['ms_per_step'])"
  head -5 $o/timeline.txt
done
