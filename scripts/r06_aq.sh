#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06aq; mkdir -p $o
for b in 32 4; do
  bash scripts/timeline.sh $b > /dev/null 2>&1
  python3 scripts/timeline.py gpurun_out/timeline/on/on_kernel_trace.csv.gz 6 -v > $o/timeline_bs$b.txt 2>&1
  cp gpurun_out/timeline/on/on_kernel_trace.csv.gz $o/trace_bs$b.csv.gz; rm -rf gpurun_out/timeline
done
head -5 $o/timeline_bs32.txt $o/timeline_bs4.txt
