#!/usr/bin/env python
"""Host enqueue time vs GPU time of one G+D step: is a small-batch step bound by the host issuing launches?
usage: host_time.py [batch ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_batch
from voicepuppet_amd.engine import PixReferEngine

def main():
  bss = [int(x) for x in sys.argv[1:]] or [4, 8, 32]
  dev = torch.device("cuda:0")
  for bs in bss:
    eng = PixReferEngine(bs, 256, 64, 64, dtype="bf16", training=True)
    eng.load_params(eng.random_params(seed=0))
    batch = synth_batch(bs, 256, 1000, dev)
    for _ in range(10):
      eng.train_step(*batch, lr=3e-4)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(30):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      eng.train_step(*batch, lr=3e-4)
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      enq.append(t1 - t0); tot.append(t2 - t0)
    enq.sort(); tot.sort()
    # back-to-back (steady state)
    t0 = time.perf_counter()
    for _ in range(50):
      eng.train_step(*batch, lr=3e-4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("bs %d: isolated step: host enqueue %.3f ms, until GPU done %.3f ms | 50 back-to-back: host %.3f ms/step, total %.3f ms/step"
          % (bs, enq[len(enq) // 2] * 1e3, tot[len(tot) // 2] * 1e3, (t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3), flush=True)
    del eng
main()
