"""debug: per-channel dbeta / dgamma of the bottleneck layers, device f32 vs the float64 oracle; repeated engines"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_step as T
o = T.oracle_step.__wrapped__() if hasattr(T.oracle_step, "__wrapped__") else None
if o is None:
  ngf = ndf = 8; n, h = 2, 256
  p = T.make_params(ngf, ndf, 3); batch = T.synth(n, h, 11)
  p64 = {k: v.astype(np.float64) for k, v in p.items()}
  st = T.ref.TrainState(p64, ngf, ndf)
  nodes = st.step(*[b.astype(np.float64) for b in batch])
  o = dict(ngf=ngf, ndf=ndf, n=n, h=h, params=p, batch=batch, nodes=nodes, after=st.p)
ref_g = o["nodes"]["Gen_grads"]
for rep in range(3):
  eng = T.run_engine(o, "f32")
  g = eng.get_params(0, src=eng.grads_g)
  for name in sorted(g):
    if "batch_normalization" not in name or "merged_" not in name: continue
    a, r = g[name].astype(np.float64), ref_g[name]
    err = np.abs(a - r)
    bad = np.where(err > 1e-4 * np.abs(r).max() + 1e-12)[0]
    if len(bad): print(rep, name, "bad", bad[:8], a[bad[:8]], r[bad[:8]])
  print(rep, "done", flush=True)
