"""Where do the three-stream and the single-stream step first differ?  python scripts/soak3.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicepuppet_amd import _lib
from voicepuppet_amd.engine import PixReferEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
L = _lib.lib()
if len(sys.argv) > 2: L.vp_tune(b"patch_min_blocks", int(sys.argv[2]))
a = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
b = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
p = a.random_params(seed=0)
a.load_params(p); b.load_params(p)
g = torch.Generator(device=dev).manual_seed(1)
batch = [torch.rand(n, 256, 256, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
L.vp_tune(b"overlap", 1); a.forward(*batch); torch.cuda.synchronize()
L.vp_tune(b"overlap", 0); b.forward(*batch); torch.cuda.synchronize()
for name in ("Outputs_raw", "gen_out4", "Outputs_FG", "Predict", "losses"):
  try:
    x, y = a.tensor(name).float(), b.tensor(name).float()
    print("fwd %-12s equal %s  max|diff| %.3e" % (name, torch.equal(x, y), float((x - y).abs().max())))
  except Exception as e:
    print("fwd", name, "n/a", e)
L.vp_tune(b"overlap", 1); a.backward(); torch.cuda.synchronize()
L.vp_tune(b"overlap", 0); b.backward(); torch.cuda.synchronize()
L.vp_tune(b"overlap", 1)
for which, (x, y) in (("grads_d", (a.grads_d, b.grads_d)), ("grads_g", (a.grads_g, b.grads_g))):
  man = a.manifests[1 if which == "grads_d" else 0]
  nd = 0
  for vn, off, shape in man:
    k = 1
    for d in shape: k *= d
    if not torch.equal(x[off:off + k], y[off:off + k]):
      nd += 1
      if nd <= 4: print("  %s: %s differs (max %.3e, |g| max %.3e)" % (which, vn, float((x[off:off+k] - y[off:off+k]).abs().max()), float(x[off:off+k].abs().max())))
  print(which, "variables that differ:", nd, "of", len(man))
