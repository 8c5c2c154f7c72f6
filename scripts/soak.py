"""Soak: N training steps on fresh random batches with the three-stream schedule and again single-stream from the same start; the
parameters must agree bit for bit (a race between streams would show as a difference or a NaN).  python scripts/soak.py [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicepuppet_amd.engine import PixReferEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
res = []
for overlap in (True, False):
  eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  eng.load_params(eng.random_params(seed=0))
  eng.profile(False)
  eng.set_option("overlap", 1 if overlap else 0)
  g = torch.Generator(device=dev).manual_seed(1)
  for s in range(steps):
    batch = [torch.rand(n, 256, 256, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
    eng.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  res.append((eng.params_g.clone(), eng.params_d.clone(), eng.losses()))
  print("overlap" if overlap else "single stream", {k: round(v, 5) for k, v in res[-1][2].items()})
ok = all(torch.isfinite(t).all().item() for r in res for t in r[:2])
same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
print("finite:", ok, " bit-identical after %d steps:" % steps, same)
sys.exit(0 if ok and same else 1)
