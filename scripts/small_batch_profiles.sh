#!/bin/bash
# Evidence of the small-batch step (the per-GPU share of the 8-GPU strong-scaling run): rocprofv3 kernel stats + a kernel timeline of the
# overlapped step at global batch 4 and 8, the HIP-event phase times, and overlapped vs single-stream step times.  (A profiled timeline
# is host-bound - rocprofv3 triples the per-launch cost - so its GAPS are not the step's; durations and the launch census are.)
# usage (via gpurun): bash scripts/small_batch_profiles.sh   ->  gpurun_out/small/
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/small
rm -rf $o; mkdir -p $o
for gb in 4 8; do
  rocprofv3 --kernel-trace --stats -d $o/p$gb -o b$gb --output-format csv -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --global-batch $gb > $o/p$gb.log 2>&1
  f=$(find $o/p$gb -name "*kernel_trace.csv" | head -1)
  python3 scripts/timeline.py $f 8 > $o/timeline_bs$gb.txt 2>&1
  python3 scripts/timeline.py $f 8 -v >> $o/timeline_bs$gb.txt 2>&1
  cp $(find $o/p$gb -name "*kernel_stats.csv" | head -1) $o/kernel_stats_bs$gb.csv
  python3 scripts/kstats_summary.py $o/kernel_stats_bs$gb.csv 12 > $o/kernel_stats_bs$gb.txt
  rm -rf $o/p$gb
done
python3 scripts/phases.py 4 8 32 2>&1 | grep batch > $o/phases.txt
bash scripts/ab.sh -b "4 8" "" "tune:overlap=0" 2>&1 | grep batch > $o/overlap_vs_single.txt
cat $o/phases.txt $o/overlap_vs_single.txt
