#!/bin/bash
# rocprofv3 kernel stats of the BFMNet training step: bash scripts/kstats_f4.sh [batch]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
b=${1:-4}
o=gpurun_out/kstats_f4_b$b
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o/r -o r --output-format csv -- python3 scripts/bench_bfmnet_train.py 20 $b 35709 ${2:-graph} > $o/r.log 2>&1
rm -f $o/r/*kernel_trace.csv
tail -2 $o/r.log
python3 - "$o/r/r_kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("kernels: %d launches, %.3f ms GPU-busy per step (25 steps incl. warm-up)" % (calls / 25, tot / 25e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
  print("%-70s calls/step %6.1f  us/step %8.1f  avg us %7.2f" % (r["Name"][:70], int(r["Calls"]) / 25, float(r["TotalDurationNs"]) / 25e3, float(r["AverageNs"]) / 1e3))
P
