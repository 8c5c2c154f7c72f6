"""Lock-step soaks of two more equalities at full width: (a) vp_pixrefer_backward_update (fused) vs backward + two Adam calls,
(b) the BFMNet training step replayed from its hipGraph vs issued eagerly (dropout off).  python scripts/soak4.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from voicepuppet_amd.engine import PixReferEngine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
ok = True
for n in (8, 32):
  a = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  b = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  p = a.random_params(seed=0)
  a.load_params(p); b.load_params(p)
  b.fused_update = False
  g = torch.Generator(device=dev).manual_seed(2)
  first = None
  for s in range(steps if n == 8 else max(steps // 4, 5)):
    batch = [torch.rand(n, 256, 256, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
    a.train_step(*batch, lr=3e-4); b.train_step(*batch, lr=3e-4); torch.cuda.synchronize()
    if not (torch.equal(a.params_g, b.params_g) and torch.equal(a.params_d, b.params_d)):
      first = s; break
  print("fused vs separate update, batch %d: %s" % (n, "identical" if first is None else "DIFFER at step %d" % first))
  ok = ok and first is None
  del a, b
from voicepuppet_amd.bfmnet.bfmnet import random_variables
from voicepuppet_amd.bfmnet.train_engine import BFMNetTrainEngine
rng = np.random.default_rng(0)
nver, B, T = 35709, 4, 24
model = {"exBase": rng.normal(0, 0.05, (3 * nver, 64)).astype(np.float32), "vmask": np.ones(3 * nver, np.float32)}
a, b = BFMNetTrainEngine(B, T, model), BFMNetTrainEngine(B, T, model)
w = random_variables(0)
a.load_params(w); b.load_params(w)
worst = 0.0
for s in range(steps):
  ears = torch.rand(B, T, 1, device=dev); mf = torch.randn(B, 5 * T, 80, device=dev); co = torch.randn(B, T, 257, device=dev) * 0.5
  ra = a.train_step(ears, mf, co, [T] * B)
  rb = b.train_step_graphed(ears, mf, co, [T] * B, 0.0, 0.0)
  worst = max(worst, float((a.arena - b.arena).abs().max()))
print("BFMNet eager vs hipGraph over %d steps: max |parameter difference| %.3e, losses %.6g / %.6g" % (steps, worst, ra["loss"], rb["loss"]))
ok = ok and worst < 1e-5 and np.isfinite(ra["loss"])
sys.exit(0 if ok else 1)
