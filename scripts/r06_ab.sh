#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
o=gpurun_out/r06ab; rm -rf $o; mkdir -p $o
for gb in 32 4; do
rocprofv3 --kernel-trace --stats -d $o/k$gb -o k --output-format csv -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary --tune streams=1 --global-batch $gb > $o/k$gb.log 2>&1
python3 scripts/kstats_summary.py $(find $o/k$gb -name "*kernel_stats.csv" | head -1) 10 2>/dev/null | head -60 > $o/kstats_$gb.txt
rm -rf $o/k$gb
done
python scripts/layer_profile.py 32 256 bf16 2>/dev/null > $o/layers32.txt
head -50 $o/kstats_32.txt
