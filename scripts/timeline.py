#!/usr/bin/env python3
"""Timeline of one overlapped training step from a rocprofv3 kernel trace (csv or csv.gz): wall time, time with >= 1 kernel
resident, idle gaps, and the per-queue picture.  usage: timeline.py trace.csv[.gz] [steps]"""
import csv, gzip, sys, re, collections

def short(n):
  n = re.sub(r"^void ", "", n)
  n = re.sub(r"\(.*", "", n)
  n = re.sub(r"vp::", "", n)
  return n[:70]

def main():
  path = sys.argv[1]
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
  f = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
  rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])) for r in csv.DictReader(f)]
  rows.sort()
  # a step starts at each pack_inputs launch
  starts = [i for i, r in enumerate(rows) if "pack_inputs" in r[3]]
  starts = starts[-steps:]
  i0, i1 = starts[-3], starts[-2]      # a steady-state step
  step = rows[i0:i1]
  t0 = step[0][0]
  wall = rows[i1][0] - t0
  print("kernels in step: %d, wall %.3f ms, sum of durations %.3f ms" % (len(step), wall / 1e6, sum(e - s for s, e, _, _ in step) / 1e6))
  # union busy
  busy = 0; cur_s, cur_e = step[0][0], step[0][1]; gaps = []
  for s, e, q, n in step[1:]:
    if s > cur_e:
      busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0, n)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
  busy += cur_e - cur_s
  print("busy (>=1 kernel) %.3f ms, idle %.3f ms in %d gaps (mean %.1f us)" % (busy / 1e6, (wall - busy) / 1e6, len(gaps), (sum(g[0] for g in gaps) / max(1, len(gaps))) / 1e3))
  # concurrency histogram
  ev = []
  for s, e, q, n in step: ev += [(s, 1), (e, -1)]
  ev.sort(); lvl = 0; last = ev[0][0]; hist = collections.Counter()
  for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
  print("concurrency (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
  perq = collections.defaultdict(float)
  for s, e, q, n in step: perq[q] += (e - s) / 1e6
  print("per queue busy ms:", dict(perq))
  if "-v" in sys.argv:
    for s, e, q, n in step: print("%9.1f %8.1f q%d %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
  else:
    print("largest gaps:")
    for g, at, n in sorted(gaps, reverse=True)[:12]: print("  %.1f us at %.2f ms before %s" % (g / 1e3, at / 1e6, n))

main()
