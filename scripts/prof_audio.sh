#!/bin/bash
# rocprofv3 kernel stats of the audio path (BASELINE config 3)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/audio
rm -rf $o; mkdir -p $o
python3 scripts/bench_audio.py 20 > $o/bench_audio.json 2> $o/bench_audio.err
rocprofv3 --kernel-trace --stats -d $o/prof -o audio --output-format csv -- python3 scripts/bench_audio.py 10 > $o/prof.log 2>&1
rm -f $o/prof/*kernel_trace.csv
cat $o/bench_audio.json; python3 scripts/kstats_summary.py $o/prof/audio_kernel_stats.csv 26 | head -30
