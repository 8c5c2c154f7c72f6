#!/bin/bash
# round 5: bfm_dwproj.hip (depthwise + projection of the inverted-residual blocks in one kernel): parity, config-3 time on / off, kernel stats
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/dwproj; mkdir -p $o
timeout 600 python -m pytest -x -q --timeout 300 tests/test_gpu_audio.py > $o/tests.log 2>&1; tail -6 $o/tests.log
for r in 1 2; do
  timeout 120 python3 scripts/bench_audio.py 30 f32 2>/dev/null | tee -a $o/bench.jsonl
  timeout 120 python3 scripts/bench_audio.py 30 f32 bfm_dwproj=0 2>/dev/null | tee -a $o/bench.jsonl
done
timeout 300 rocprofv3 --kernel-trace --stats -d $o/prof -o audio --output-format csv -- python3 scripts/bench_audio.py 10 > $o/prof.log 2>&1
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 && cp "$f" $o/kernel_stats.csv
rm -rf $o/prof
