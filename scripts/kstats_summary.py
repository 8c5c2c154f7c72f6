#!/usr/bin/env python3
"""per-step totals from a rocprofv3 kernel_stats.csv: usage kstats_summary.py stats.csv <steps incl. warmup>"""
import csv, re, sys
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
out = []
for r in rows:
  n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Name"])).replace("vp::", "")
  ms = float(r["TotalDurationNs"]) / 1e6 / steps
  out.append((ms, int(r["Calls"]) / steps, n[:90]))
  tot += ms
conv = sum(m for m, c, n in out if re.search(r"igemm|wgrad_kernel|wgrad_tr|conv_cin8|deconv_cout4", n))
print("total %.3f ms/step, conv %.3f, other %.3f, launches/step %.1f" % (tot, conv, tot - conv, sum(c for m, c, n in out)))
for m, c, n in sorted(out, reverse=True)[:45]:
  print("%8.3f ms %6.1f calls  %s" % (m, c, n))
