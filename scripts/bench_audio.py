"""BASELINE.json config 3 alone (log-mel -> BFMNet, bs = 64 x 1 s of 16 kHz audio, f32) so that rocprofv3 can profile just the audio path:
python scripts/bench_audio.py [steps [f32|bf16 [key=value ...]]] (key=value: vp_tune knobs, e.g. bfm_dwproj=0).  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import audio_ref as ar   # pcm length helper + BFMNet initialiser only (scripts/ is not the product path)
from voicepuppet_amd.audio import LogMel, BFMNetEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
trunk = sys.argv[2] if len(sys.argv) > 2 else "f32"      # "bf16": MfccNet activations / 1x1-conv operands in bf16
tunes = sys.argv[3:]
if tunes:
  from voicepuppet_amd import _lib
  for kv in tunes:
    k, v = kv.split("=")
    _lib.check(_lib.lib().vp_tune(k.encode(), int(v)), "vp_tune " + kv)

def timed(fn, warm, n):
  for _ in range(warm): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

B, T = 64, 25
pcm = torch.tensor(np.random.default_rng(0).normal(0, 0.1, (B, ar.pcm_length_for(T))).astype(np.float32), device="cuda")
lm = LogMel(B, pcm.shape[1]); net = BFMNetEngine(B, T, dtype=trunk); net.load_params(ar.init_bfmnet_params(0, dtype=np.float32))
ears = torch.full((B, T, 1), 0.3, device="cuda"); seq = [T] * B
dt_lm = timed(lambda: lm(pcm), 3, steps)
mf = lm(pcm)
dt_net = timed(lambda: net.forward(ears, mf, seq), 3, steps)
print(json.dumps({"config": "log-mel -> BFMNet %s bs=64 x 1 s%s" % (trunk, (" [" + " ".join(tunes) + "]") if tunes else ""), "logmel_ms": dt_lm * 1e3, "bfmnet_ms": dt_net * 1e3,
                  "audio_seconds_per_s": B / (dt_lm + dt_net), "logmel_GBps": 4 * (pcm.numel() + mf.numel()) / dt_lm / 1e9,
                  "bfmnet_tflops": 10.64e9 * B / dt_net / 1e12}))
