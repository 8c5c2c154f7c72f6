"""Per-layer conv kernel timing (HIP events) of one G+D step: python scripts/layer_profile.py [batch] [height] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from voicepuppet_amd.engine import PixReferEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
for kv in sys.argv[4:]:          # library knobs (before the plan is made), e.g. tune:thin_blocks_cout4=1024
  k, v = kv.split("=")
  if k.startswith("tune:"):
    from voicepuppet_amd import _lib
    _lib.check(_lib.lib().vp_tune(k[5:].encode(), int(v)))
eng = PixReferEngine(n, h, 64, 64, dtype=dt, training=True)
eng.load_params(eng.random_params(0))
for kv in sys.argv[4:]:          # plan options, e.g. bwd_sums_in_epilogue=0
  k, v = kv.split("=")
  if not k.startswith("tune:"):
    eng.set_option(k, int(v))
batch = bench.synth_batch(n, h, 1, torch.device("cuda"))
for _ in range(2): eng.train_step(*batch, lr=3e-4)
torch.cuda.synchronize()
eng.profile(2)
steps = 3
for _ in range(steps): eng.train_step(*batch, lr=3e-4)
torch.cuda.synchronize()
recs = eng.profile_collect(); eng.profile(0)
recs.sort(key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in recs) / steps
print("conv total %.3f ms/step" % tot)
for r in recs:
  ms = r["ms"] / steps
  print("%-52s %8.3f ms %8.1f TF %8.1f GB/s" % (r["name"], ms, r["flops"] / steps / ms / 1e9, r["bytes"] / steps / ms / 1e6))
