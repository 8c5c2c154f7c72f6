"""Three consecutive steps, oracle (float64) vs engine (f32), small width: where does step >= 1 diverge?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pixrefer_ref as ref
from voicepuppet_amd.engine import PixReferEngine
ngf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
s = np.load(os.path.join(G, "sample22_256.npz"))
frame, face3d, matte = [s[k].astype(np.float32) / 255.0 for k in ("frame", "face3d", "matte")]
inputs = np.concatenate([face3d, face3d], axis=-1)[None].repeat(n, 0)
fg = np.concatenate([frame * matte, frame * matte], axis=-1)[None].repeat(n, 0)
batch = [inputs, fg, frame[None].repeat(n, 0), matte[None].repeat(n, 0)]
if len(sys.argv) > 3:   # random batch instead
  rng = np.random.default_rng(11)
  batch = [rng.uniform(size=(n, 256, 256, c)).astype(np.float32) for c in (6, 6, 3, 3)]
p = ref.init_params(ngf, ngf, seed=9, dtype=np.float32)
st = ref.TrainState({k: v.astype(np.float64) for k, v in p.items()}, ngf, ngf)
eng = PixReferEngine(n, 256, ngf, ngf, dtype="f32", training=True)
eng.load_params(p)
dev = [torch.tensor(b, device="cuda") for b in batch]
for step in range(3):
  nodes = st.step(*[b.astype(np.float64) for b in batch])
  eng.forward(*dev); eng.backward(); torch.cuda.synchronize()
  got = eng.losses()
  print("step", step, {k: "%.2e" % (abs(got[k] - nodes[k]) / abs(nodes[k])) for k in got})
  pr = eng.tensor("Predict").cpu().numpy()
  print("   predict_real relL2 %.2e  predict_fake relL2 %.2e" % (
      np.linalg.norm(pr[0].ravel() - nodes["Predict_real"].ravel()) / np.linalg.norm(nodes["Predict_real"]),
      np.linalg.norm(pr[1].ravel() - nodes["Predict_fake"].ravel()) / np.linalg.norm(nodes["Predict_fake"])))
  gd = eng.get_params(1, src=eng.grads_d)
  worst = sorted(((np.linalg.norm(gd[k] - nodes["Discrim_grads"][k]) / max(np.linalg.norm(nodes["Discrim_grads"][k]), 1e-30), k) for k in gd), reverse=True)[:3]
  print("   worst D grads", [("%.1e" % a, b) for a, b in worst])
  eng.adam_step(3e-4); torch.cuda.synchronize()
  now = dict(eng.get_params(0), **eng.get_params(1))
  worst = sorted(((np.linalg.norm(now[k] - st.p[k]) / max(np.linalg.norm(st.p[k] - p[k].astype(np.float64)), 1e-30), k) for k in now), reverse=True)[:6]
  print("   worst params after the update (error / size of total update)", [("%.1e" % a, b.split('/', 1)[1]) for a, b in worst])
