"""Algorithmic bytes of the hand-written kernels of one BFMNet training step vs their measured time:
   python scripts/f4_roofline.py <r_kernel_stats.csv> <steps in the trace> <batch> [nver] > profiles/..._roofline.json
Bytes are counted from the layer shapes (tinynet.py:172-203): every tensor a kernel must read or write once, float32."""
import csv
import json
import sys

BLOCKS = [(64, 1, False), (64, 6, True), (64, 6, False), (128, 6, True), (128, 6, False), (128, 6, False), (192, 6, True), (192, 6, False),
          (192, 6, False), (192, 6, False), (256, 6, False), (256, 6, False), (256, 6, False), (256, 6, True), (256, 6, False), (256, 6, False),
          (256, 6, False)]
PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def layers(B, T=24):
  """[(pixels, channels, kind)] of every batch-normalised tensor; kind: 'conv' (1x1 / stem / block8) or 'dw'."""
  H, W = 5 * T, 40
  out = [(B * H * W, 32, "conv", True)]
  cin = 32
  for cout, exp, pool in BLOCKS:
    P = B * H * W
    out.append((P, cin * exp, "conv", True))       # expansion + relu6
    out.append((P, cin * exp, "dw", True))         # depthwise + relu6
    out.append((P, cout, "conv", False))           # projection
    if cout != cin:
      out.append((P, cout, "conv", False))         # shortcut
    cin = cout
    if pool:
      W = -(-W // 2)
  out.append((B * H * W, 256, "conv", True))
  return out


def gemm_flops(B, T=24, nver=35709):
  """forward + data gradient + weight gradient of every 1x1 convolution / the stem, and the three face-shape products"""
  H, W = 5 * T, 40
  fl = 3 * 2 * B * H * W * 48 * 32
  cin = 32
  for cout, exp, pool in BLOCKS:
    P = B * H * W
    fl += 3 * 2 * P * (cin * cin * exp + cin * exp * cout + (cin * cout if cout != cin else 0))
    cin = cout
    if pool:
      W = -(-W // 2)
  return fl + 3 * 2 * B * H * W * 256 * 256 + 3 * 2 * B * T * 64 * 3 * nver


def main():
  path, steps, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
  L = layers(B)
  el = lambda sel: sum(p * c for p, c, k, a in L if sel(k, a))
  every, dws, acts = el(lambda k, a: True), el(lambda k, a: k == "dw"), el(lambda k, a: a)
  nbn = len(L)
  algo = {   # elements moved per step, by kernel
    "chan_sums_kernel<0>": every,                 # read y
    "chan_sums_kernel<1>": 2 * every,             # read y, da
    "bn_bwd_apply2_kernel": 3 * every,            # read y, da; write dx
    "affine_act_kernel": 2 * every,               # read y, write a   (the five dense-layer calls are negligible)
    "dwconv7x3_kernel": 2 * 2 * dws,              # forward + backward-data: read, write
    "dwconv7x3_wgrad_kernel": 2 * dws,            # read a, dy
  }
  rows = list(csv.DictReader(open(path)))
  res = {"batch": B, "frames": 24, "batch_norm_layers": nbn, "peak_GBps": PEAK, "kernels": []}
  tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps
  res["gpu_busy_ms_per_step"] = tot / 1e6
  res["launches_per_step"] = sum(int(r["Calls"]) for r in rows) / steps
  gemm = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("Cijk")) / steps
  res["rocblas_gemm_ms_per_step"] = gemm / 1e6
  res["rocblas_gemm_GFLOP_per_step"] = gemm_flops(B) / 1e9
  res["rocblas_gemm_TFLOPs_f32"] = gemm_flops(B) / gemm / 1e3
  res["f32_mfma_peak_TFLOPs"] = 157.3
  for r in rows:
    for k, elems in algo.items():
      if k in r["Name"]:
        us = float(r["TotalDurationNs"]) / steps / 1e3
        gb = elems * 4 / 1e9
        res["kernels"].append({"kernel": k, "calls_per_step": int(r["Calls"]) / steps, "us_per_step": us, "avg_us": float(r["AverageNs"]) / 1e3,
                               "algorithmic_GB_per_step": gb, "achieved_GBps": gb / (us / 1e6), "frac_of_hbm_peak": gb / (us / 1e6) / PEAK})
  res["kernels"].sort(key=lambda d: -d["us_per_step"])
  print(json.dumps(res, indent=1))


if __name__ == "__main__":
  main()
