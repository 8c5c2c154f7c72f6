#!/bin/bash
# Collects the round's judged evidence on the GPU box into gpurun_out/<tag>_*: the bench line, the rocprofv3 kernel
# stats of the same command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only with --kernel-trace).
# usage (via gpurun):  bash scripts/profile_round.sh r01e
tag=${1:-r01}
out=gpurun_out
mkdir -p $out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python3 bench.py --steps 30 --warmup 10 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
# kernel stats twice: the step as it runs (three streams: co-running kernels stretch each other) and single-stream
# (--tune streams=1: the per-kernel durations bench.py's HIP-event pass measures - it also runs on one stream)
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o ${tag} --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary > $out/${tag}_prof.log 2>&1
ONE="--tune streams=1"
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof1 -o ${tag}_single --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary $ONE > $out/${tag}_prof1.log 2>&1
# the counter passes keep the single-stream schedule ON PURPOSE: bench.py's roofline object times the dominant class in its single-stream
# HIP-event pass (full-batch launches), and `traffic` must be bytes of THOSE launches, not of the half-batch launches of the
# three-stream schedule.  (The step total therefore does not contain the pool-only saving of the overlapped schedule, 0.4 GB.)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary $ONE > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary $ONE > $out/${tag}_pmc_write.log 2>&1
python3 scripts/pmc_summary.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc > $out/${tag}_pmc_summary.txt 2>&1
# keep the merge-back small: the raw traces are large
rm -f $out/${tag}_prof/*kernel_trace.csv $out/${tag}_prof1/*kernel_trace.csv $out/${tag}_pmc_fetch/*kernel_trace.csv $out/${tag}_pmc_write/*kernel_trace.csv $out/${tag}_pmc_fetch/*counter_collection.csv $out/${tag}_pmc_write/*counter_collection.csv
tail -c 600 $out/${tag}_bench.json; cat $out/${tag}_pmc_summary.txt | head -8
