#!/usr/bin/env python
"""Critical-path time of every generator layer (forward and backward) of the overlapped step, from HIP events in front of each
layer on the caller's stream (vp_tune("phase_marks", 2)); no profiler.  usage: layer_chain.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import synth_batch
from voicepuppet_amd.engine import PixReferEngine
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng = PixReferEngine(bs, 256, 64, 64, dtype="bf16", training=True)
eng.load_params(eng.random_params(seed=0))
batch = synth_batch(bs, 256, 1000, torch.device("cuda:0"))
eng.fused_update = False
for _ in range(10): eng.train_step(*batch, lr=3e-4)
names = ["encoder_1", "encoder_2", "encoder_3", "encoder_4", "encoder_fg_1", "encoder_fg_2", "encoder_fg_3", "encoder_fg_4", "merged_encoder_2",
         "merged_encoder_3", "merged_encoder_4", "merged_encoder_5", "merged_decoder_5", "merged_decoder_4", "merged_decoder_3", "merged_decoder_2",
         "merged2_decoder_4", "merged2_decoder_3", "merged2_decoder_2", "decoder_1"]
eng.L.vp_tune(b"phase_marks", 2)
F, B = [], []
main = [i for i, n in enumerate(names) if not n.startswith("encoder_fg")]
for _ in range(20):
  torch.cuda.synchronize(); eng.train_step(*batch, lr=3e-4); torch.cuda.synchronize()
  f = [eng.L.vp_pixrefer_mark_ms(eng.h, 8 + a, 8 + b) for a, b in zip(main[:-1], main[1:])] + [eng.L.vp_pixrefer_mark_ms(eng.h, 8 + main[-1], 1)]
  rb = main[::-1]
  b = [eng.L.vp_pixrefer_mark_ms(eng.h, 32 + a, 32 + b2) for a, b2 in zip(rb[:-1], rb[1:])] + [eng.L.vp_pixrefer_mark_ms(eng.h, 32 + rb[-1], 6)]
  F.append(f); B.append(b)
eng.L.vp_tune(b"phase_marks", 0)
F, B = np.median(np.array(F), 0) * 1e3, np.median(np.array(B), 0) * 1e3
print("batch %d: generator forward chain %.0f us, backward chain %.0f us (per layer, on the caller's stream)" % (bs, F.sum(), B.sum()))
for k, i in enumerate(main): print("  fwd %-20s %6.1f us" % (names[i], F[k]))
for k, i in enumerate(main[::-1]): print("  bwd %-20s %6.1f us" % (names[i], B[k]))
