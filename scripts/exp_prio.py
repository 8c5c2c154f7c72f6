"""Experiment: the step's main chain on a HIGH-priority stream (its small dependent kernels then win free CU slots against the
side / branch streams' big convolutions).  python scripts/exp_prio.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicepuppet_amd.engine import PixReferEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
h = 256
dev = torch.device("cuda", 0)
eng = PixReferEngine(n, h, 64, 64, dtype="bf16", training=True)
eng.load_params(eng.random_params(seed=0))
g = torch.Generator(device=dev).manual_seed(0)
batch = [torch.rand(n, h, h, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)

def timed(fn, steps=40, warm=10):
  for _ in range(warm): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps * 1e3

for rep in range(2):
  base = timed(lambda: eng.train_step(*batch, lr=3e-4))
  for prio in (-1, 0):
    s = torch.cuda.Stream(priority=prio)
    def step():
      s.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(s):
        eng.train_step(*batch, lr=3e-4)
      torch.cuda.current_stream().wait_stream(s)
    print("bs%d default stream %.3f ms | own stream priority %d: %.3f ms" % (n, base, prio, timed(step)))
