"""Inference fidelity of frame batching at full width: N frames in one launch with per-sample batch-norm statistics vs the
reference's N separate batch-1 runs (infer_bfmvid.py:238-243).  python scripts/check_frame_batch.py [frames] [height] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicepuppet_amd.engine import PixReferEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
g = torch.Generator(device="cuda").manual_seed(0)
x = [torch.rand(n, h, h, c, device="cuda", generator=g) for c in (6, 3, 3)]
eb = PixReferEngine(n, h, 64, 64, dtype=dt, training=False, per_sample_bn=True)
p = eb.random_params(seed=0)
eb.load_params(p)
eb.forward(*x)
got = eb.tensor("Outputs_raw").float().clone()
e1 = PixReferEngine(1, h, 64, 64, dtype=dt, training=False)
e1.load_params(p)
worst, same = 0.0, 0
for i in range(n):
  e1.forward(*[t[i:i + 1].contiguous() for t in x])
  one = e1.tensor("Outputs_raw")[0].float()
  same += int(torch.equal(got[i], one))
  worst = max(worst, float((got[i] - one).norm() / one.norm()))
print("%s %dx%d: %d frames batched vs one by one: %d bit-identical, worst rel-L2 %.3e" % (dt, h, h, n, same, worst))
