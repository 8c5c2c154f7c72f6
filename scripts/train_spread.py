#!/usr/bin/env python
"""how far may the bf16 trajectory sit from the float32 one?  As far as float32 trajectories sit from EACH OTHER.

200 iterations of the reference's schedule on the four-sample full-width fixture (scripts/train_curves.py), for three seeds of the
initial weights, on: the float32 engine, the float32 engine again with every initial weight perturbed by one float32 ulp-scale factor
(1 + 1e-6 * N(0,1): the float32 path's own chaos), the bf16 engine, the bf16 engine with vp_tune("patch_min_blocks", 256) (another kernel
class for several small-batch layers = another K-sum order: the arm that left the 5 % band of round 5).

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

AT = (10, 50, 100, 200)


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
    L.vp_tune(b"patch_min_blocks", 256)
    try:
      runs[(s, "bf16_pmb256")] = tc.run_curves("bf16", steps, p, batch)[0]
    finally:
      L.vp_tune(b"patch_min_blocks", 384)
  return runs


def distances(runs, seeds, steps):
  """per checkpoint and loss key: (seed-to-seed spread of float32, worst float32-vs-perturbed, worst bf16-vs-float32 over seeds and arms);
  relative to the float32 window mean for the smooth terms, absolute (nats) for the two GAN terms"""
  K = {k: i for i, k in enumerate(tc.KEYS)}
  out = {}
  for at in AT:
    a = min(at, steps - 1)
    w = {key: tc.window_mean(c, a) for key, c in runs.items()}
    row = {}
    for k, i in K.items():
      rel = k in ("Gen_loss_L1", "Perceptual_loss", "Gen_loss")
      f = np.array([w[(s, "f32")][i] for s in seeds])
      scale = f.mean() if rel else 1.0
      spread = (f.max() - f.min()) / scale
      chaos = max(abs(w[(s, "f32_perturbed")][i] - w[(s, "f32")][i]) for s in seeds) / scale
      low = max(abs(w[(s, arm)][i] - w[(s, "f32")][i]) for s in seeds for arm in ("bf16", "bf16_pmb256")) / scale
      row[k] = {"f32_seed_spread": float(spread), "f32_vs_perturbed_f32": float(chaos), "bf16_vs_f32": float(low), "relative": rel,
                "f32_means": [float(x) for x in f]}
    out[at] = row
  return out


if __name__ == "__main__":
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "train_spread.json")
  seeds = (0, 1, 2)
  runs = run_all(steps, seeds)
  dist = distances(runs, seeds, steps)
  rec = {"steps": steps, "keys": tc.KEYS, "seeds": list(seeds), "checkpoints": AT, "distances": dist,
         "curves": {"%d/%s" % k: v.tolist() for k, v in runs.items()}}
  os.makedirs(os.path.dirname(path), exist_ok=True)
  json.dump(rec, open(path, "w"))
  for at in AT:
    print("step %d" % at)
    for k, r in dist[at].items():
      print("  %-16s f32 seed-to-seed %.4f | f32 vs perturbed f32 %.4f | bf16 vs f32 %.4f  (%s; f32 means %s)" %
            (k, r["f32_seed_spread"], r["f32_vs_perturbed_f32"], r["bf16_vs_f32"], "relative" if r["relative"] else "nats",
             " ".join("%.4f" % x for x in r["f32_means"])))
