#!/bin/bash
# round-6 evidence on one box: the judged profile set (bench line with its sub-records, kernel stats four-stream + single-stream, FETCH /
# WRITE PMC passes), matrix-pipe counters, per-layer conv times, phase times, audio path, A/B against the end-of-round-5 library,
# the atomic-contention probe.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r06_ev; rm -rf $o; mkdir -p $o
timeout 1500 bash scripts/profile_round.sh r06 > $o/profile_round.log 2>&1
timeout 900 bash scripts/pmc_mix.sh > /dev/null 2>&1; cp gpurun_out/pmc_mix/mix.txt $o/pmc_instruction_mix.txt; cp gpurun_out/pmc_mix/mfma_busy.json $o/pmc_mfma_busy.json
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layer_times.txt 2>&1
timeout 300 python scripts/phases.py 32 8 4 > $o/phases.txt 2>&1
python3 scripts/bench_audio.py 20 f32 > $o/audio_bench.json 2> $o/audio.err
python3 scripts/bench_audio.py 20 bf16 >> $o/audio_bench.json 2>> $o/audio.err
timeout 300 rocprofv3 --kernel-trace --stats -d $o/aprof -o audio --output-format csv -- python3 scripts/bench_audio.py 10 > $o/aprof.log 2>&1
f=$(find $o/aprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/audio_kernel_stats.csv; rm -rf $o/aprof
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" "tune:streams=1" > $o/ab_vs_r5.txt 2>&1
scripts/probes/atomic_probe > $o/atomic_probe.txt 2>&1
python scripts/exp_host_bound.py 4 8 32 2>&1 | grep -v amdgpu.ids > $o/exp_host_enqueue_rate.txt
for b in 32 4; do
  bash scripts/timeline.sh $b > /dev/null 2>&1
  python3 scripts/timeline.py gpurun_out/timeline/on/on_kernel_trace.csv.gz 6 -v > $o/timeline_bs$b.txt 2>&1
  rm -rf gpurun_out/timeline
done
tail -c 400 gpurun_out/r06_bench.json; grep "^batch" $o/ab_vs_r5.txt; cat $o/phases.txt $o/atomic_probe.txt
