#!/usr/bin/env python
"""Folds two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs of the same bench command, as
MI355X_MICROARCH.md 'HBM' prescribes) into per-kernel HBM bytes per launch:

    hbm_bytes = 2 * FETCH_SIZE[KB] * 1024  +  WRITE_SIZE[KB] * 1024        (gfx950: FETCH_SIZE under-reports by 2x)

usage: pmc_summary.py <fetch_dir> <write_dir> <out_prefix>     -> <out_prefix>_hbm_traffic.json, _fetch_per_kernel.csv, _write_per_kernel.csv
Kernel classes are decoded from the template arguments so bench.py can look a tile shape up by name."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

IGEMM_TILES = {(2, 2, 4, 4): "128x128", (1, 4, 4, 2): "64x128", (1, 4, 1, 2): "16x128", (4, 1, 2, 2): "128x32", (4, 1, 2, 1): "128x16",
               (2, 2, 2, 1): "64x32", (2, 4, 4, 4): "128x256", (2, 4, 8, 4): "256x256", (1, 4, 4, 4): "64x256"}
WGRAD_TILES = {(2, 2, 4, 4): "128x128", (2, 2, 4, 2): "128x64", (4, 1, 2, 1): "128x16"}


def classify(name):
  """'igemm_bf16_128x256' style class of a vp:: conv kernel, from either the mangled name or rocprofv3's (sometimes
  garbled: bf16 'DF16b' confuses its demangler and swallows the first int) demangled one; None for other kernels."""
  if "conv_c64_kernel" in name:     # conv_c64_kernel<NCH, TPW, NW, REF, RELU, POOL, STORE>: 16 TPW NW output channels, 4 x 16-pixel tiles
    m = re.search(r"conv_c64_kernel<(\d+), (\d+), (\d+)", name) or re.search(r"conv_c64_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name)
    return "c64_bf16_%dx64" % (16 * int(m.group(2)) * int(m.group(3)) if m else 64)
  if "conv_dc64_kernel" in name:
    return "dc64_bf16_64x128"
  if "conv_s2c64_kernel" in name:
    return "s2c64_bf16_128x64"
  if "conv_cin8_kernel" in name:
    return "cin8_bf16_64x16"
  if "deconv_cout4" in name:
    return "cout4_bf16_16x16"
  if "deconv_cout8_tile_kernel" in name:
    return "dcout8_bf16_32x64"
  if "conv3x3_cout8_tile_kernel" in name:
    return "cout8_bf16_16x64"
  if "wgrad_tr_kernel" in name:   # wgrad_tr_kernel<WM, WN, TC, TP, NST, FAST, EXACT>: class = variant + tile (rows = WM*TC*16, columns = WN*TP*16)
    m = re.search(r"wgrad_tr_kernel<(.*?)>", name)
    if not m:
      return "wgrad_tr_bf16_256x128"
    args = [x.strip() for x in m.group(1).split(",")]
    ints = [int(x) for x in args if x.isdigit()]
    flags = [x == "true" for x in args if x in ("true", "false")]
    wm, wn, tc, tp = ints[:4]
    var = "tr_exact" if len(flags) > 1 and flags[1] else ("tr_fast" if flags and flags[0] else "tr")
    return "wgrad_%s_bf16_%dx%d" % (var, wm * tc * 16, wn * tp * 16)
  for fam in ("igemm_patch3_kernel", "igemm_patch2_kernel", "igemm_patch_kernel"):
    if fam not in name:
      continue
    m = re.search(fam + r"I(DF16b|f)((?:Li\d+E)+)", name)
    if m:
      dt = "bf16" if m.group(1) == "DF16b" else "f32"
      ints = [int(x) for x in re.findall(r"Li(\d+)E", m.group(2))]
    else:
      m = re.search(fam + r"<(.*?)>\(", name)
      if not m:
        return None
      args = [a.strip() for a in m.group(1).split(",")]
      dt = "f32" if args[0] == "float" else "bf16"
      ints = [int(a) for a in args if a.isdigit()]
    if len(ints) < 6:
      return None
    wc, wp, tc, tp, th, tw = ints[:6]
    return "%s_%s_%dx%d" % ({"igemm_patch3_kernel": "patch3", "igemm_patch2_kernel": "patch2", "igemm_patch_kernel": "patch"}[fam], dt, wc * tc * 16, th * tw)
  for fam, tiles in (("igemm_dma_kernel", IGEMM_TILES), ("igemm_ws_kernel", IGEMM_TILES), ("igemm_regb_kernel", IGEMM_TILES), ("wgrad_kernel", WGRAD_TILES)):
    if fam not in name:
      continue
    m = re.search(fam + r"I(DF16b|f)((?:Li\d+E)+)", name)
    if m:
      dt = "bf16" if m.group(1) == "DF16b" else "f32"
      ints = tuple(int(x) for x in re.findall(r"Li(\d+)E", m.group(2)))[:4]
    else:
      m = re.search(fam + r"<(.*?)>\(", name)
      if not m:
        return None
      args = [a.strip() for a in m.group(1).split(",")]
      nums = [int(a) for a in args if a.isdigit()]
      dt = "f32" if args[0] == "float" else "bf16"
      want = 5 if fam in ("wgrad_kernel", "igemm_ws_kernel") else 4
      ints = tuple(nums[:4]) if len(nums) >= want else tuple([1] + nums[:3])     # garbled form lost WC=1
    t = tiles.get(ints)
    if not t:
      return None
    if fam.startswith("igemm"):      # one class per kernel template, as bench.py's HIP-event records name them
      return "igemm_%s_%s_%s" % ({"igemm_dma_kernel": "dma", "igemm_ws_kernel": "ws", "igemm_regb_kernel": "regb"}[fam], dt, t)
    return "wgrad_%s_%s" % (dt, t)
  return None


def load(d, counter):
  rows = defaultdict(lambda: [0, 0.0, 0.0])
  for f in glob.glob(d + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
      if r["Counter_Name"] != counter:
        continue
      e = rows[r["Kernel_Name"]]
      e[0] += 1
      e[1] += float(r["Counter_Value"])
      e[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
  return rows


def main():
  fdir, wdir, prefix = sys.argv[1:4]
  fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
  out = []
  for k, (n, kb, us) in fetch.items():
    wn, wkb, _ = write.get(k, [0, 0.0, 0.0])
    if "vp::" not in k and "_ZN2vp" not in k:
      continue
    f, w = kb / n, (wkb / wn if wn else 0.0)
    out.append({"kernel": k, "class": classify(k), "launches": n, "fetch_kb_raw": f, "write_kb": w,
                "hbm_bytes_per_launch": (2 * f + w) * 1024, "avg_us_under_pmc": us / n})
  out.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
  json.dump(out, open(prefix + "_hbm_traffic.json", "w"), indent=1)
  for tag, rows in (("fetch", fetch), ("write", write)):
    with open("%s_%s_per_kernel.csv" % (prefix, tag), "w") as fo:
      cw = csv.writer(fo)
      cw.writerow(["kernel", "launches", "avg_kb_per_launch", "avg_us"])
      for k, (n, kb, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        cw.writerow([k, n, "%.3f" % (kb / n), "%.3f" % (us / n)])
  for r in out[:12]:
    print("%-22s n=%4d hbm %.1f MB/launch  %.1f us" % (r["class"] or r["kernel"][:22], r["launches"], r["hbm_bytes_per_launch"] / 1e6, r["avg_us_under_pmc"]))


if __name__ == "__main__":
  main()
