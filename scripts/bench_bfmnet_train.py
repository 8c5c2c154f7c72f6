"""ms per BFMNet training step on one MI355X: `python scripts/bench_bfmnet_train.py [steps] [batch] [nver] [eager|graph|auto] [one]` (one: the eager step on one stream).
Synthetic clips of 24 frames (the generator's slice length), a random stand-in face model of `nver` vertices (35709 = BFM_model_front),
dropout on, loss fetched every step as train_bfmnet.py does.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicepuppet_amd.bfmnet.train_engine import BFMNetTrainEngine


def main():
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
  B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
  nver = int(sys.argv[3]) if len(sys.argv) > 3 else 35709
  mode = sys.argv[4] if len(sys.argv) > 4 else "graph"
  T = 24
  rng = np.random.default_rng(0)
  vm = np.ones((nver, 3), np.float32)
  vm[rng.choice(nver, nver // 20, replace=False)] = 10
  eng = BFMNetTrainEngine(B, T, {"exBase": rng.normal(0, 0.05, (3 * nver, 64)).astype(np.float32), "vmask": vm.reshape(-1)},
                          side_stream=not (len(sys.argv) > 5 and sys.argv[5] == "one"))
  from voicepuppet_amd.bfmnet.bfmnet import random_variables
  eng.load_params(random_variables(0))
  dev = eng.dev
  ears = torch.rand(B, T, 1, device=dev)
  mfccs = torch.randn(B, 5 * T, 80, device=dev)
  coeff = torch.randn(B, T, 257, device=dev) * 0.5
  seq = torch.full((B,), T, dtype=torch.int32, device=dev)
  if mode == "auto":
    step = lambda: eng.train_step_auto(ears, mfccs, coeff, seq, 0.25)
  elif mode == "graph":
    step = lambda: eng.train_step_graphed(ears, mfccs, coeff, seq, 0.25)
  else:
    step = lambda: eng.train_step(ears, mfccs, coeff, seq, masks=eng.draw_masks(0.25))
  for _ in range(5):
    step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    r = step()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / steps * 1e3
  print(json.dumps({"metric": "bfmnet_train_clips_per_sec", "value": B / ms * 1e3, "unit": "clips/s", "ms_per_step": ms, "steps": steps,
                    "config": {"workload": "BFMNet build_train_op", "batch": B, "frames": T, "vertices": nver, "dropout": True, "mode": mode},
                    "dtype": "f32", "loss": r["loss"], "global_norm": r["global_norm"]}))


if __name__ == "__main__":
  main()
