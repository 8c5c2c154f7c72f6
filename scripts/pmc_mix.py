"""per-kernel instruction mix (+ matrix-pipe busy fraction) from rocprofv3 --pmc dirs:
python scripts/pmc_mix.py <mix dir> [<busy dir> [<out.json>]]
mix dir: SQ_INSTS_VALU / _MFMA / _SALU / _LDS; busy dir: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES (+ SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY).
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8): the SQ counter sums the matrix-pipe busy cycles of all 1024 SIMDs
(MI355X_MICROARCH.md: it counts cycles, 16 per v_mfma_f32_16x16x32_bf16); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs
(GUI_ACTIVE / duration = 19-21 "GHz" on every long kernel), one eighth of it is the cycles the kernel kept the GPU busy at the clock it
actually ran at - so the fraction is against the matrix peak AT THAT CLOCK (the counter passes run at 1.9-2.1 GHz: DVFS), not against
2.4 GHz.  mfma_issue_2p4 = 16 * SQ_INSTS_MFMA / (1024 * 2.4 GHz * kernel time): the same from instruction counts at the nominal clock."""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import classify
from collections import defaultdict


def read(d):
  acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int); dur = defaultdict(float)
  seen = set()
  for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
      n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("vp::", "")[:80]
      acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
      did = r["Dispatch_Id"]
      if did not in seen:
        seen.add(did); calls[n] += 1
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
  return acc, calls, dur


acc, calls, dur = read(sys.argv[1])
busy = {}
if len(sys.argv) > 2:
  bacc, bcalls, bdur = read(sys.argv[2])
  for k, a in bacc.items():
    if a.get("GRBM_GUI_ACTIVE", 0) > 0:
      busy[k] = {"class": classify(k), "mfma_busy": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * a["GRBM_GUI_ACTIVE"] / 8.0),
                 "mfma_busy_cycles": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), "gui_active": a["GRBM_GUI_ACTIVE"], "busy_cu_cycles": a.get("SQ_BUSY_CU_CYCLES", 0),
                 "effective_clock_ghz": a["GRBM_GUI_ACTIVE"] / 8.0 / (bdur[k] * 1e6) if bdur[k] > 0 else None,
                 "wait_inst_frac": a.get("SQ_WAIT_INST_ANY", 0) / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES", 0) > 0 else None,
                 "calls": bcalls[k], "ms": bdur[k]}
print("%-82s %6s %8s %10s %8s %8s %8s %9s %9s" % ("kernel", "calls", "ms", "MFMA(M)", "VALU/M", "SALU/M", "LDS/M", "mfma_busy", "issue@2.4"))
for k in sorted(acc, key=lambda k: -dur[k]):
  a = acc[k]; m = a.get("SQ_INSTS_MFMA", 0)
  if m <= 0: continue
  b = busy.get(k, {}).get("mfma_busy")
  issue = 16.0 * m / (1024.0 * 2.4e9 * dur[k] * 1e-3)
  if k in busy:
    busy[k]["mfma_issue_2p4"] = issue
  print("%-82s %6d %8.3f %10.2f %8.2f %8.2f %8.2f %9s %9.3f" % (k, calls[k], dur[k], m / 1e6, a.get("SQ_INSTS_VALU", 0) / m, a.get("SQ_INSTS_SALU", 0) / m,
                                                               a.get("SQ_INSTS_LDS", 0) / m, "%.3f" % b if b is not None else "-", issue))
if len(sys.argv) > 3:
  json.dump(busy, open(sys.argv[3], "w"), indent=1)
