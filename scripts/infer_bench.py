"""Generator-only inference latency: python scripts/infer_bench.py [batch] [height]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from voicepuppet_amd.engine import PixReferEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng = PixReferEngine(n, h, 64, 64, dtype="bf16", training=False, per_sample_bn=True)
eng.load_params(eng.random_params(0))
b = bench.synth_batch(n, h, 1, torch.device("cuda"))
for _ in range(5): eng.forward(b[0], b[1], b[2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): eng.forward(b[0], b[1], b[2])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("inference bs=%d %dx%d: %.3f ms  %s" % (n, h, h, dt * 1e3, {k: v for k, v in os.environ.items() if k.startswith("VP_")}))
# the same forward as one hipGraph launch (the library's launch sequence is fixed and never synchronises)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
  eng.forward(b[0], b[1], b[2])
  torch.cuda.synchronize()
  with torch.cuda.graph(g, stream=s):
    eng.forward(b[0], b[1], b[2])
torch.cuda.synchronize()
ref_out = eng.tensor("Outputs_raw").clone()
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("  as hipGraph: %.3f ms, output identical: %s" % (dt * 1e3, bool(torch.equal(ref_out, eng.tensor("Outputs_raw")))))
