"""A/B of the generator-only inference forward (per-sample batch-norm, bf16) under VP_LIB: python scripts/infer_ab.py [n] [h]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from voicepuppet_amd.engine import PixReferEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng = PixReferEngine(n, h, 64, 64, dtype="bf16", training=False, per_sample_bn=True); eng.load_params(eng.random_params(0))
b = bench.synth_batch(n, h, 1, torch.device("cuda"))
for _ in range(60): eng.forward(b[0], b[1], b[2])
res = []
for r in range(3):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(50): eng.forward(b[0], b[1], b[2])
  torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50 * 1e3)
print("inference bs=%d %dx%d [%s]: %s ms" % (n, h, h, os.path.basename(os.environ.get("VP_LIB", "libvp_hip.so")), " ".join("%.3f" % x for x in res)))
