"""per-kernel instruction mix from a rocprofv3 --pmc dir: python scripts/pmc_mix.py <dir>"""
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int); dur = defaultdict(float)
seen = set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("vp::", "")[:80]
    key = n
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    did = r["Dispatch_Id"]
    if did not in seen:
      seen.add(did); calls[key] += 1
      dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("%-82s %6s %8s %10s %8s %8s %8s" % ("kernel", "calls", "ms", "MFMA(M)", "VALU/M", "SALU/M", "LDS/M"))
for k in sorted(acc, key=lambda k: -dur[k]):
  a = acc[k]; m = a.get("SQ_INSTS_MFMA", 0)
  if m <= 0: continue
  print("%-82s %6d %8.3f %10.2f %8.2f %8.2f %8.2f" % (k, calls[k], dur[k], m / 1e6, a.get("SQ_INSTS_VALU", 0) / m, a.get("SQ_INSTS_SALU", 0) / m, a.get("SQ_INSTS_LDS", 0) / m))
