#!/usr/bin/env python
"""Phase times of the G+D step (HIP events on the main stream, no profiler): usage phases.py [batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import synth_batch
from voicepuppet_amd.engine import PixReferEngine
NAMES = ["G fwd", "D/VGG fwd+loss", "D(G)/VGG bwd+comp (+call gap)", "G bwd stage0", "G bwd stage1", "G bwd stage2+joins", "join D pass"]
for bs in [int(x) for x in sys.argv[1:]] or [4, 8, 32]:
  eng = PixReferEngine(bs, 256, 64, 64, dtype="bf16", training=True)
  eng.load_params(eng.random_params(seed=0))
  batch = synth_batch(bs, 256, 1000, torch.device("cuda:0"))
  eng.fused_update = False          # forward / backward / adam as separate calls: the marks live in forward and backward
  for _ in range(10): eng.train_step(*batch, lr=3e-4)
  eng.L.vp_tune(b"phase_marks", 1)
  acc = []
  for _ in range(20):
    torch.cuda.synchronize()
    eng.train_step(*batch, lr=3e-4)
    torch.cuda.synchronize()
    acc.append(eng.phase_ms())
  eng.L.vp_tune(b"phase_marks", 0)
  m = np.median(np.array(acc), axis=0)
  print("batch %d: total %.3f ms | " % (bs, m.sum()) + ", ".join("%s %.3f" % (n, v) for n, v in zip(NAMES, m)), flush=True)
  del eng
