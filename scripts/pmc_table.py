"""Per-kernel table of the counters in a rocprofv3 --pmc output dir: python scripts/pmc_table.py <dir> [name filter]"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else "igemm"
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int)); dur = defaultdict(float)
for f in glob.glob(d + "/*counter_collection.csv"):
  for r in csv.DictReader(open(f)):
    if filt not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"][:60], r["Grid_Size"])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
    dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for key in acc:
  print(key[0], "grid", key[1], "last_us %.1f" % dur[key])
  for c in sorted(acc[key]):
    print("   %-40s %16.1f  (n=%d)" % (c, acc[key][c] / cnt[key][c], cnt[key][c]))
