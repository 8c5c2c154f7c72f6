#!/bin/bash
# kernel timelines: the bs-32 step (which stream carries what), and a multi-stream hipGraph replay at bs 4 (root cause of the 2.3x)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
o=gpurun_out/r06b; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace -d $o/t32 -o t32 --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary > $o/t32.log 2>&1
f=$(find $o/t32 -name "*kernel_trace.csv" | head -1); python3 scripts/timeline.py $f 6 -v > $o/timeline_bs32.txt 2>&1; head -8 $o/timeline_bs32.txt
rocprofv3 --kernel-trace -d $o/g4 -o g4 --output-format csv -- python3 scripts/exp_graph2.py 4 tracemulti > $o/g4.log 2>&1
f=$(find $o/g4 -name "*kernel_trace.csv" | head -1); python3 scripts/timeline.py $f 6 -v > $o/timeline_graph_multi_bs4.txt 2>&1; head -8 $o/timeline_graph_multi_bs4.txt
rocprofv3 --kernel-trace -d $o/e4 -o e4 --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary --global-batch 4 > $o/e4.log 2>&1
f=$(find $o/e4 -name "*kernel_trace.csv" | head -1); python3 scripts/timeline.py $f 6 -v > $o/timeline_eager_bs4.txt 2>&1; head -8 $o/timeline_eager_bs4.txt
find $o -name "*kernel_trace.csv" -exec gzip -9 {} \;
du -sh $o
