#!/usr/bin/env python
"""Loss trajectories of the float32 and the bf16 PixReferNet engine from the SAME initial weights on the four-sample full-width
fixture (tests/golden/full_width_n4.npz, ngf = ndf = 64, 256x256), the reference's schedule (train_pixrefer.py:134-143: Adam(D) then
Adam(G) per iteration, lr = 3e-4 * 0.999^floor(global_step / 1000), global_step += 2).

  python scripts/train_curves.py [steps] [out.json]

`run_curves` is what tests/test_gpu_training_trajectory.py asserts on; this script writes the two curves for profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

KEYS = ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss")


def learning_rate(base, global_step, decay_steps=1000, decay_rate=0.999):
  """tf.train.exponential_decay(staircase=True), pixrefer.py:391-394."""
  return base * decay_rate ** (global_step // decay_steps)


def fixture_batch():
  d = np.load(os.path.join(ROOT, "tests", "golden", "full_width_n4.npz"))
  batch = [d[k].astype(np.float32) / 255.0 for k in ("inputs", "fg_inputs", "targets", "masks")]
  return int(d["ngf"]), int(d["seed"]), batch


def initial_params(seed):
  """The reference's initialisers (pixrefer.py:64,68,100-101), drawn once on the host: both engines start from the same arrays."""
  from voicepuppet_amd.engine import PixReferEngine
  eng = PixReferEngine(4, 256, 64, 64, dtype="f32", training=True)
  p = eng.random_params(seed)
  del eng
  return p


def run_curves(dtype, steps, params, batch, fused=True):
  """[steps, 5] losses of the forward pass of every step (before that step's update), and the generator's output after the last."""
  from voicepuppet_amd.engine import PixReferEngine
  eng = PixReferEngine(4, 256, 64, 64, dtype=dtype, training=True)
  eng.load_params(params)
  eng.fused_update = fused
  dev = [torch.tensor(b, device="cuda") for b in batch]
  out = np.zeros((steps, len(KEYS)), np.float64)
  for s in range(steps):
    eng.train_step(*dev, lr=learning_rate(3e-4, 2 * s))
    got = eng.losses()                   # (reads the loss buffer: waits for the step)
    out[s] = [got[k] for k in KEYS]
  eng.forward(*dev)
  torch.cuda.synchronize()
  pix = ((eng.tensor("Outputs_raw") + 1) / 2).float().cpu().numpy()
  del eng
  return out, pix


def window_mean(curve, at, half=5):
  lo, hi = max(0, at - half), min(len(curve), at + half)
  return curve[lo:hi].mean(axis=0)


if __name__ == "__main__":
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "train_curves.json")
  ngf, seed, batch = fixture_batch()
  params = initial_params(seed)
  rec = {"steps": steps, "keys": KEYS, "fixture": "tests/golden/full_width_n4.npz", "ngf": ngf}
  pix = {}
  for dt in ("f32", "bf16"):
    c, pix[dt] = run_curves(dt, steps, params, batch)
    rec[dt] = c.tolist()
  rec["final_pixels_rel_l2_bf16_vs_f32"] = float(np.linalg.norm(pix["bf16"] - pix["f32"]) / np.linalg.norm(pix["f32"]))
  rec["final_l1_to_target"] = {dt: float(np.abs(pix[dt] - batch[2]).mean()) for dt in pix}
  os.makedirs(os.path.dirname(path), exist_ok=True)
  json.dump(rec, open(path, "w"))
  f, b = np.array(rec["f32"]), np.array(rec["bf16"])
  print("step  " + "  ".join("%-26s" % k for k in KEYS))
  for at in [0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 300, 400, 500]:
    if at > steps:
      break
    a = min(at, steps - 1)
    wf, wb = window_mean(f, a), window_mean(b, a)
    print("%4d  " % at + "  ".join("%10.4f /%10.4f    " % (x, y) for x, y in zip(wf, wb)))
  print("final pixels bf16 vs f32 rel-L2 %.3e; mean |Outputs - targets| f32 %.4f bf16 %.4f" %
        (rec["final_pixels_rel_l2_bf16_vs_f32"], rec["final_l1_to_target"]["f32"], rec["final_l1_to_target"]["bf16"]))
