"""Is the few-frame step bound by the host's launch rate?  Enqueue K steps without synchronising and compare the host time of the
enqueue loop with the time to completion:  python scripts/exp_host_bound.py [BATCH ...]
(host-bound: the two agree; device-bound: the enqueue loop returns early - but note HIP queues are finite: once ~1000s of packets
are outstanding the launch call blocks, so K is kept small)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from voicepuppet_amd.engine import PixReferEngine

for n in [int(a) for a in sys.argv[1:]] or [4, 8, 32]:
  eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  eng.load_params(eng.random_params(seed=0))
  batch = bench.synth_batch(n, 256, 1, torch.device("cuda"))
  for _ in range(10): eng.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  for K in (1, 2, 4):
    best = None
    for rep in range(5):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(K): eng.train_step(*batch, lr=3e-4)
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      r = ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3)
      best = r if best is None or r[1] < best[1] else best
    print("batch %d, %d steps enqueued back to back: host enqueue %.3f ms per step, completion %.3f ms per step" % (n, K, best[0], best[1]), flush=True)
  del eng
