#!/bin/bash
# rocprofv3 kernel stats of the bench step (no CPU baseline, no f32, no event profiling), overlap on and off
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/kstats
mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o/on -o on --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline > $o/on.log 2>&1
VP_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d $o/off -o off --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline > $o/off.log 2>&1
rm -f $o/on/*kernel_trace.csv $o/off/*kernel_trace.csv
tail -c 300 $o/on.log; ls $o/on $o/off
