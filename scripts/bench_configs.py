"""Secondary BASELINE.json configs on one GPU: (2) G+D step bs=8 256x256 bf16, (4, per GPU) bs=8 512x512 bf16,
(3) log-mel -> BFMNet bs=64 x 1 s, and generator-only inference fps.  Prints one JSON object per config."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import audio_ref as ar   # pcm length helper + BFMNet initialiser only (scripts/ is not the product path)
from voicepuppet_amd.engine import PixReferEngine
from voicepuppet_amd.audio import LogMel, BFMNetEngine

def timed(fn, warm, steps):
  for _ in range(warm): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

for n, h in ((8, 256), (8, 512), (2, 512)):
  eng = PixReferEngine(n, h, 64, 64, dtype="bf16", training=True); eng.load_params(eng.random_params(0))
  b = bench.synth_batch(n, h, 1, torch.device("cuda"))
  dt = timed(lambda: eng.train_step(*b, lr=3e-4), 10, 20)
  print(json.dumps({"config": "G+D step bf16 bs=%d %dx%d 1 GPU" % (n, h, h), "ms_per_step": dt * 1e3, "frames_per_s": n / dt,
                    "tflops": 163.02e9 * (h / 256) ** 2 * n / dt / 1e12}))
  del eng; torch.cuda.empty_cache()
for n, h in ((1, 512), (8, 512), (1, 256)):
  eng = PixReferEngine(n, h, 64, 64, dtype="bf16", training=False, per_sample_bn=True); eng.load_params(eng.random_params(0))
  b = bench.synth_batch(n, h, 1, torch.device("cuda"))
  dt = timed(lambda: eng.forward(b[0], b[1], b[2]), 50, 30)      # (the first calls of a process load the code objects of kernels no earlier config used: a one-off of milliseconds)
  print(json.dumps({"config": "generator inference bf16 bs=%d %dx%d" % (n, h, h), "ms": dt * 1e3, "frames_per_s": n / dt}))
  del eng; torch.cuda.empty_cache()
B, T = 64, 25
pcm = torch.tensor(np.random.default_rng(0).normal(0, 0.1, (B, ar.pcm_length_for(T))).astype(np.float32), device="cuda")
lm = LogMel(B, pcm.shape[1]); net = BFMNetEngine(B, T); net.load_params(ar.init_bfmnet_params(0, dtype=np.float32))
ears = torch.full((B, T, 1), 0.3, device="cuda"); seq = [T] * B
dt_lm = timed(lambda: lm(pcm), 3, 20)
mf = lm(pcm)
dt_net = timed(lambda: net.forward(ears, mf, seq), 3, 10)
print(json.dumps({"config": "log-mel -> BFMNet f32 bs=64 x 1 s", "logmel_ms": dt_lm * 1e3, "bfmnet_ms": dt_net * 1e3,
                  "audio_seconds_per_s": B / (dt_lm + dt_net), "logmel_GBps": 4 * (pcm.numel() + mf.numel()) / dt_lm / 1e9,
                  "bfmnet_tflops": 10.64e9 * B / dt_net / 1e12}))
