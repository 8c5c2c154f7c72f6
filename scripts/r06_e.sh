#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
o=gpurun_out/r06e; mkdir -p $o
python scripts/layer_profile.py 32 256 bf16 bwd_sums_in_epilogue=1 2>/dev/null > $o/layers_on.txt
python scripts/layer_profile.py 32 256 bf16 bwd_sums_in_epilogue=0 2>/dev/null > $o/layers_off.txt
python - <<'P'
a={l.split()[0]:float(l.split()[-6]) for l in open('gpurun_out/r06e/layers_on.txt') if ' ms ' in l and ':' in l.split()[0]}
b={l.split()[0]:float(l.split()[-6]) for l in open('gpurun_out/r06e/layers_off.txt') if ' ms ' in l and ':' in l.split()[0]}
tot=0
for k in sorted(a, key=lambda k:-(a[k]-b.get(k,a[k]))):
  d=a[k]-b.get(k,a[k])
  if abs(d)>0.002: print("%-50s on %.3f off %.3f  diff %+.3f"%(k,a[k],b.get(k,0),d)); tot+=d
print("sum of diffs", tot)
P
# kernel stats single stream on/off for the non-conv kernels
for v in 1 0; do
rocprofv3 --kernel-trace --stats -d $o/k$v -o k --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary --tune streams=1 --tune bwd_sums_in_epilogue=$v > $o/k$v.log 2>&1
python3 scripts/kstats_summary.py $(find $o/k$v -name "*kernel_stats.csv" | head -1) 7 2>/dev/null | head -70 > $o/kstats_$v.txt
rm -rf $o/k$v
done
head -5 $o/kstats_1.txt
