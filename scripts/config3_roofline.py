#!/usr/bin/env python
"""BASELINE config 3 (log-mel -> BFMNet, 64 clips x 1 s, float32) against its MIXED per-launch roofline, the way SURVEY.md 8d prices the
PixReferNet step: every launch of the plan (plan_bfmnet.hip: stem, 17 inverted-residual blocks = expansion 1x1 GEMM -> [depthwise 7x3 +
projection 1x1 GEMM in one kernel] (+ shortcut GEMM), pools, block8, head) costs max(flops / 157.3 TF, bytes / 6.3 TB/s) with every
tensor read / written once per launch (the expanded tensor therefore twice: written by the expansion, read by the fused depthwise +
projection kernel); the table says which blocks are bound by what.  No GPU: python scripts/config3_roofline.py [measured_ms]"""
import sys

B, T5, W0 = 64, 125, 80
PEAK, HBM = 157.3e12, 6.3e12
specs = [("block1_0", 64, 1, False), ("block2_0", 64, 6, True), ("block2_1", 64, 6, False), ("block3_0", 128, 6, True), ("block3_1", 128, 6, False),
         ("block3_2", 128, 6, False), ("block4_0", 192, 6, True), ("block4_1", 192, 6, False), ("block4_2", 192, 6, False), ("block4_3", 192, 6, False),
         ("block5_0", 256, 6, False), ("block5_1", 256, 6, False), ("block5_2", 256, 6, False), ("block6_0", 256, 6, True), ("block6_1", 256, 6, False),
         ("block6_2", 256, 6, False), ("block7_0", 256, 6, False)]     # tinynet.py:159-212 (scope, cout, expansion, pool after)


def t(flops, byts):
  return max(flops / PEAK, byts / HBM)


rows = []
W = 40                                 # block0: conv [9,5] stride [1,2] on 80 mel bins
P = B * T5 * W
rows.append(("block0_0 stem 9x5", 2.0 * P * 45 * 32, 4.0 * (B * T5 * W0 + P * 32)))
cin = 32
tot_exp_bytes = 0.0
for scope, cout, e, pool in specs:
  P = B * T5 * W
  cexp = cin * e
  rows.append((scope + " expansion %d->%d (W=%d)" % (cin, cexp, W), 2.0 * P * cin * cexp, 4.0 * P * (cin + cexp)))
  rows.append((scope + " depthwise+projection %d->%d" % (cexp, cout), 2.0 * P * cexp * (21 + cout), 4.0 * P * (cexp + cout + (cout if cout == cin else 0))))
  tot_exp_bytes += 2 * 4.0 * P * cexp
  if cout != cin:
    rows.append((scope + " shortcut %d->%d" % (cin, cout), 2.0 * P * cin * cout, 4.0 * P * (cin + cout + cout)))
  if pool:
    Wn = (W + 1) // 2
    rows.append((scope + " max-pool", 0.0, 4.0 * (P * cout + B * T5 * Wn * cout)))
    W = Wn
  cin = cout
P = B * T5 * W
rows.append(("block8_0 1x1 256->256", 2.0 * P * 256 * 256, 4.0 * P * 512))
BT = B * 25
head = 2.0 * BT * (256 * 256 * 2 + 256 * 512 + 256 * 256 + 512 * 512 / 2 + 256 * 128 + 128 * 64 + 64 * 64)
rows.append(("head (pool 5x3, dense, GRU, decoder)", head, 4.0 * (P * 256 + BT * 2048)))

tot_f = sum(r[1] for r in rows)
tot_b = sum(r[2] for r in rows)
tot_t = sum(t(r[1], r[2]) for r in rows)
print("%-52s %9s %9s %8s %8s  %s" % ("launch", "GFLOP", "MB", "mfma us", "hbm us", "bound"))
for name, f, b in rows:
  print("%-52s %9.2f %9.1f %8.1f %8.1f  %s" % (name, f / 1e9, b / 1e6, f / PEAK * 1e6, b / HBM * 1e6, "mfma" if f / PEAK >= b / HBM else "hbm"))
print("total: %.1f GFLOP, %.2f GB (of which the 6x-expanded tensors, written once and read once: %.2f GB)" % (tot_f / 1e9, tot_b / 1e9, tot_exp_bytes / 1e9))
print("pure-MFMA bound %.2f ms, pure-HBM bound %.2f ms, mixed per-launch roofline %.2f ms" % (tot_f / PEAK * 1e3, tot_b / HBM * 1e3, tot_t * 1e3))
print("with the expanded tensors never leaving the chip (expansion fused too): bytes %.2f GB -> HBM bound %.2f ms" %
      ((tot_b - tot_exp_bytes) / 1e9, (tot_b - tot_exp_bytes) / HBM * 1e3))
if len(sys.argv) > 1:
  ms = float(sys.argv[1])
  print("measured %.2f ms: %.1f TF = %.2f of the float32 MFMA peak; %.2f of the mixed roofline" % (ms, tot_f / ms / 1e9, tot_f / ms / 1e9 / 157.3, tot_t * 1e3 / ms))
