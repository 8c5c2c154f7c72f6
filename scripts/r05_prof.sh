#!/bin/bash
# phase times + overlapped timeline + single-stream kernel stats of the current build (bs 32)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r05_prof; mkdir -p $o
timeout 300 python scripts/phases.py 32 > $o/phases.txt 2>&1; tail -1 $o/phases.txt
bash scripts/timeline.sh 32 > /dev/null 2>&1
python3 scripts/timeline.py gpurun_out/timeline/on/on_kernel_trace.csv.gz > $o/timeline.txt 2>&1; head -6 $o/timeline.txt
rm -rf gpurun_out/timeline
rocprofv3 --kernel-trace --stats -d $o/prof1 -o s --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --tune streams=1 > $o/prof1.log 2>&1
cp $o/prof1/s_kernel_stats.csv $o/single_stream_kernel_stats.csv; rm -rf $o/prof1
python3 - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05_prof/single_stream_kernel_stats.csv')))
for r in rows[:48]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/7:6.1f} {float(r['TotalDurationNs'])/7e6:8.3f} {float(r['AverageNs'])/1e3:8.1f}")
P
