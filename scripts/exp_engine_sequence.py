"""Why is the 4-frame step sometimes slow inside the full bench.py sequence (2.8 - 4.5 ms) and never in a fresh process (2.10 - 2.15)?
Times a 4-frame engine (a) first thing in the process, (b) after a 32-frame engine was created, run and destroyed, (c) after a float32
engine as well, (d) again; optionally keeping the earlier engines alive.   python scripts/exp_engine_sequence.py [keep]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from voicepuppet_amd.engine import PixReferEngine

keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
held = []


def timed(n, dtype="bf16", steps=40, warm=10):
  eng = PixReferEngine(n, 256, 64, 64, dtype=dtype, training=True)
  eng.load_params(eng.random_params(seed=0))
  batch = bench.synth_batch(n, 256, 1, torch.device("cuda"))
  for _ in range(warm): eng.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps): eng.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / steps * 1e3
  if keep: held.append(eng)
  else:
    del eng
    gc.collect(); torch.cuda.empty_cache()
  return ms


print("keep earlier engines alive:", keep)
print("4 frames, first engine of the process: %.3f ms" % timed(4), flush=True)
print("32 frames: %.3f ms" % timed(32), flush=True)
print("4 frames after it: %.3f ms" % timed(4), flush=True)
print("32 frames float32: %.3f ms" % timed(32, "f32", 8, 2), flush=True)
print("4 frames after it: %.3f ms" % timed(4), flush=True)
print("4 frames again: %.3f ms" % timed(4), flush=True)
print("8 frames: %.3f ms" % timed(8), flush=True)
print("4 frames again: %.3f ms" % timed(4), flush=True)
