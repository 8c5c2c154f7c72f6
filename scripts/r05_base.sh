#!/bin/bash
# round-5 baseline (the end-of-round-4 library on today's box): step times, which layers still run a separate statistics pass,
# per-layer conv times, the overlapped step's timeline
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r05_base; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" > $o/ab.txt 2>&1
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layer_times.txt 2>&1
timeout 300 python scripts/phases.py > $o/phases.txt 2>&1
bash scripts/timeline.sh 32 > /dev/null 2>&1
python3 scripts/timeline.py gpurun_out/timeline/on/on_kernel_trace.csv.gz > $o/timeline.txt 2>&1
cp gpurun_out/timeline/on/on_kernel_trace.csv.gz $o/trace_bs32.csv.gz; rm -rf gpurun_out/timeline
cat $o/ab.txt; head -30 $o/timeline.txt
