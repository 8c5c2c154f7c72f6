#!/usr/bin/env python
"""how far may the bf16 trajectory sit from the float32 one?  As far as float32 trajectories sit from EACH OTHER.
(python scripts/train_spread.py --recompute in.json out.json: the statistics again from stored curves, no GPU.)

200 iterations of the reference's schedule on the four-sample full-width fixture (scripts/train_curves.py), for three seeds of the
initial weights, on: the float32 engine, the float32 engine again with every initial weight perturbed by one float32 ulp-scale factor
(1 + 1e-6 * N(0,1): the float32 path's own chaos), the bf16 engine, the bf16 engine under the plan heuristics of rounds 2-5
(vp_tune "patch_min_blocks" 384, "igemm_splitk_target" 128: other kernel classes and K splits for several small-batch layers = other K-sum
orders; round 5's version of this arm left its 5 % band).

  python scripts/train_spread.py [steps] [out.json]

Prints, per checkpoint, the window means and three distances: seed-to-seed spread of the float32 runs, float32 vs perturbed float32
(same seed), bf16 (both arms) vs float32 (same seed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np

import train_curves as tc

AT = (50, 100, 150)          # window centres
HALF = 25                    # ... of 50-iteration windows: per-step losses of this GAN carry 10-20 % bumps that last 2-3 iterations and come at
                             # different iterations in every run (a 10-step window at the knee of the curve holds one or not: the same bf16 build
                             # measured 5 % and 10 % from float32 at step 50 with two summation orders of ONE reduction); 50 steps average them out


def window_stat(curve, at, rel, half=HALF):
  """geometric mean (the decaying terms) / arithmetic mean (the GAN terms, nats) of curve[at - half, at + half)"""
  seg = np.asarray(curve)[max(0, at - half):at + half]
  return float(np.exp(np.log(seg).mean())) if rel else float(seg.mean())


def perturbed(params, seed, eps=1e-6):
  rng = np.random.default_rng(1000 + seed)
  return {k: (v * (1.0 + eps * rng.standard_normal(v.shape))).astype(np.float32) for k, v in params.items()}


def run_all(steps, seeds):
  from voicepuppet_amd import _lib
  L = _lib.lib()
  ngf, seed0, batch = tc.fixture_batch()
  runs = {}
  for s in seeds:
    p = tc.initial_params(seed0 + s)
    runs[(s, "f32")] = tc.run_curves("f32", steps, p, batch)[0]
    runs[(s, "f32_perturbed")] = tc.run_curves("f32", steps, perturbed(p, s), batch)[0]
    runs[(s, "bf16")] = tc.run_curves("bf16", steps, p, batch)[0]
    L.vp_tune(b"patch_min_blocks", 384)            # the round-2..5 plan heuristics: other kernel classes / K splits for several layers
    L.vp_tune(b"igemm_splitk_target", 128)
    try:
      runs[(s, "bf16_alt")] = tc.run_curves("bf16", steps, p, batch)[0]
    finally:
      L.vp_tune(b"patch_min_blocks", -1)
      L.vp_tune(b"igemm_splitk_target", -1)
  return runs


REL_KEYS = ("Gen_loss_L1", "Perceptual_loss", "Gen_loss")


def distances(runs, seeds, steps):
  """per window and loss key: seed-to-seed spread of float32, worst float32-vs-perturbed-float32 (same seed), worst bf16-vs-float32 (same
  seed, both arms); relative to the float32 window value for the decaying terms, absolute (nats) for the two GAN terms"""
  K = {k: i for i, k in enumerate(tc.KEYS)}
  out = {}
  for at in AT:
    row = {}
    for k, i in K.items():
      rel = k in REL_KEYS
      w = {key: window_stat(c[:, i], at, rel) for key, c in runs.items()}
      f = np.array([w[(s, "f32")] for s in seeds])
      scale = f.mean() if rel else 1.0
      spread = (f.max() - f.min()) / scale
      chaos = max(abs(w[(s, "f32_perturbed")] - w[(s, "f32")]) / (w[(s, "f32")] if rel else 1.0) for s in seeds)
      low = max(abs(w[(s, arm)] - w[(s, "f32")]) / (w[(s, "f32")] if rel else 1.0) for s in seeds for arm in ("bf16", "bf16_alt"))
      row[k] = {"f32_seed_spread": float(spread), "f32_vs_perturbed_f32": float(chaos), "bf16_vs_f32": float(low), "relative": rel,
                "f32_values": [float(x) for x in f]}
    out[at] = row
  return out


def report(dist):
  lines = []
  for at in AT:
    lines.append("iterations [%d, %d)" % (at - HALF, at + HALF))
    for k, r in dist[at].items():
      lines.append("  %-16s f32 seed-to-seed %.4f | f32 vs 1e-6-perturbed f32 %.4f | bf16 (3 seeds x 2 arms) vs f32 %.4f  (%s; f32 %s)" %
                   (k, r["f32_seed_spread"], r["f32_vs_perturbed_f32"], r["bf16_vs_f32"], "relative" if r["relative"] else "nats",
                    " ".join("%.4f" % x for x in r["f32_values"])))
  return "\n".join(lines)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--recompute":
  rec = json.load(open(sys.argv[2]))
  seeds = tuple(rec["seeds"])
  runs = {(int(k.split("/")[0]), k.split("/")[1]): np.array(v) for k, v in rec["curves"].items()}
  rec["distances"] = distances(runs, seeds, rec["steps"])
  rec["windows"] = {"centres": AT, "half": HALF}
  json.dump(rec, open(sys.argv[3], "w"))
  print(report(rec["distances"]))
elif __name__ == "__main__":
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "train_spread.json")
  seeds = (0, 1, 2)
  runs = run_all(steps, seeds)
  dist = distances(runs, seeds, steps)
  rec = {"steps": steps, "keys": tc.KEYS, "seeds": list(seeds), "windows": {"centres": AT, "half": HALF}, "distances": dist,
         "curves": {"%d/%s" % k: v.tolist() for k, v in runs.items()}}
  os.makedirs(os.path.dirname(path), exist_ok=True)
  json.dump(rec, open(path, "w"))
  print(report(dist))
