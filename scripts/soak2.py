"""Lock-step soak: two engines from the same start, one with the three-stream schedule, one single-stream (the knob is process-wide,
so it is flipped around each call); after every step the gradient and parameter arenas are compared bit for bit.
python scripts/soak2.py [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicepuppet_amd import _lib
from voicepuppet_amd.engine import PixReferEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
L = _lib.lib()
a = PixReferEngine(n, H, 64, 64, dtype=(sys.argv[4] if len(sys.argv) > 4 else "bf16"), training=True)
b = PixReferEngine(n, H, 64, 64, dtype=(sys.argv[4] if len(sys.argv) > 4 else "bf16"), training=True)
p = a.random_params(seed=0)
a.load_params(p); b.load_params(p)
g = torch.Generator(device=dev).manual_seed(1)
b.set_option("overlap", 0)      # per handle: `a` keeps the multi-stream schedule
bad = 0
for s in range(steps):
  batch = [torch.rand(n, H, H, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
  a.train_step(*batch, lr=3e-4); b.train_step(*batch, lr=3e-4); torch.cuda.synchronize()
  for name, x, y in (("grads_g", a.grads_g, b.grads_g), ("grads_d", a.grads_d, b.grads_d), ("params_g", a.params_g, b.params_g), ("params_d", a.params_d, b.params_d)):
    if not torch.equal(x, y):
      idx = (x != y).nonzero().flatten()
      # which variable
      which = None
      man = a.manifests[0 if name.endswith("_g") else 1]
      for vn, off, shape in man:
        k = 1
        for d in shape: k *= d
        if off <= int(idx[0]) < off + k: which = vn
      print("step %d: %s differs at %d elements, first in %s (|diff| max %.3e)" % (s, name, idx.numel(), which, float((x - y).abs().max())))
      bad += 1
      break
  if bad:
    break
print("no difference in %d steps" % steps if not bad else "DIFFERENCE")
