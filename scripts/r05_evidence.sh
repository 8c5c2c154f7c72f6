#!/bin/bash
# round-5 evidence on one box: the judged profile set (bench line, kernel stats three-stream + single-stream, FETCH / WRITE PMC passes),
# the instruction mix + matrix-pipe busy counters, per-layer conv times, phase times, secondary configs, audio path, the small-batch
# experiments (hipGraph replay of the step; single stream), A/B against the end-of-round-4 library.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r05_ev; rm -rf $o; mkdir -p $o
timeout 1500 bash scripts/profile_round.sh r05 > $o/profile_round.log 2>&1
timeout 900 bash scripts/pmc_mix.sh > /dev/null 2>&1; cp gpurun_out/pmc_mix/mix.txt $o/pmc_instruction_mix.txt; cp gpurun_out/pmc_mix/mfma_busy.json $o/pmc_mfma_busy.json
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layer_times.txt 2>&1
timeout 300 python scripts/phases.py 32 8 4 > $o/phases.txt 2>&1
timeout 900 python scripts/bench_configs.py > $o/secondary_configs.jsonl 2> $o/secondary.err
python3 scripts/bench_audio.py 20 f32 > $o/audio_bench.json 2> $o/audio.err
python3 scripts/bench_audio.py 20 bf16 >> $o/audio_bench.json 2>> $o/audio.err
python3 scripts/bench_audio.py 20 f32 bfm_dwproj=0 >> $o/audio_bench.json 2>> $o/audio.err
timeout 300 rocprofv3 --kernel-trace --stats -d $o/aprof -o audio --output-format csv -- python3 scripts/bench_audio.py 10 > $o/aprof.log 2>&1
f=$(find $o/aprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/audio_kernel_stats.csv; rm -rf $o/aprof
for b in 4 8; do timeout 300 python scripts/exp_graph.py $b 256 2>&1 | tail -2; done > $o/exp_graph.txt 2>&1
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r4.so" "tune:streams=1" > $o/ab_vs_r4.txt 2>&1
tail -c 400 gpurun_out/r05_bench.json; head -12 $o/pmc_instruction_mix.txt; cat $o/ab_vs_r4.txt $o/exp_graph.txt $o/phases.txt
