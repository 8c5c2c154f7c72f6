#!/bin/bash
# the secondary evidence of a round: per-layer times, secondary configs, audio path stats, SQ counters of the isolated convs,
# instruction mix of the step, kernel timeline.  usage (via gpurun): bash scripts/evidence_round.sh
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/evidence
rm -rf $o; mkdir -p $o
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layer_times.txt 2>&1
timeout 900 python scripts/bench_configs.py > $o/secondary_configs.jsonl 2> $o/secondary.err
python3 scripts/bench_audio.py 20 f32 > $o/audio_bench.json 2> $o/audio.err
python3 scripts/bench_audio.py 20 bf16 >> $o/audio_bench.json 2>> $o/audio.err
rocprofv3 --kernel-trace --stats -d $o/aprof -o audio --output-format csv -- python3 scripts/bench_audio.py 10 > $o/aprof.log 2>&1
cp $o/aprof/audio_kernel_stats.csv $o/audio_kernel_stats.csv; rm -rf $o/aprof
bash scripts/pmc_sq.sh ev > /dev/null 2>&1; cp gpurun_out/ev_all.txt $o/pmc_sq_isolated_convs.txt
bash scripts/pmc_mix.sh > /dev/null 2>&1; cp gpurun_out/pmc_mix/mix.txt $o/pmc_instruction_mix.txt
bash scripts/timeline.sh 32 > /dev/null 2>&1
python3 scripts/timeline.py gpurun_out/timeline/on/on_kernel_trace.csv.gz > $o/timeline.txt 2>&1
ls -la $o; cat $o/secondary_configs.jsonl; cat $o/timeline.txt | head -5
