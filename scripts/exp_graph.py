"""Experiment: the whole PixReferNet G+D step replayed from a hipGraph vs issued eagerly (timing only: the captured Adam step count
is frozen).  python scripts/exp_graph.py [batch] [height]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicepuppet_amd.engine import PixReferEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
eng = PixReferEngine(n, h, 64, 64, dtype="bf16", training=True)
eng.load_params(eng.random_params(seed=0))
g = torch.Generator(device=dev).manual_seed(0)
batch = [torch.rand(n, h, h, c, device=dev, generator=g) for c in (6, 6, 3, 3)]

def timed(fn, steps=30, warm=10):
  for _ in range(warm): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps * 1e3

eager = timed(lambda: eng.train_step(*batch, lr=3e-4))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
  for _ in range(3): eng.train_step(*batch, lr=3e-4)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
  eng.train_step(*batch, lr=3e-4)
rep = timed(graph.replay)
print("bs%d %dx%d: eager %.3f ms, hipGraph replay %.3f ms" % (n, h, h, eager, rep))
