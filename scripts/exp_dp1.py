"""Experiment: the data-parallel code path (staged generator backward, bucketed RCCL all-reduces on the collective stream) on ONE GPU
with a one-rank RCCL group that really executes the collectives; compares with the plain single-GPU step.  python scripts/exp_dp1.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
from voicepuppet_amd import parallel
from voicepuppet_amd.engine import PixReferEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
three = len(sys.argv) > 2 and sys.argv[2] == "three"      # the executor drops to three streams under the data-parallel schedule (rounds 3-5; round 6 keeps four)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
eng.load_params(eng.random_params(seed=0))
if three:
  eng.use_streams(3)
g = torch.Generator(device=dev).manual_seed(0)
batch = [torch.rand(n, 256, 256, c, device=dev, generator=g) for c in (6, 6, 3, 3)]

def timed(fn, steps=30, warm=10):
  for _ in range(warm): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps * 1e3

before = timed(lambda: eng.train_step(*batch, lr=3e-4))
dist.init_process_group("nccl", device_id=dev)
t = torch.ones(1 << 20, device=dev); dist.all_reduce(t); torch.cuda.synchronize()      # communicator + its streams exist now
plain = timed(lambda: eng.train_step(*batch, lr=3e-4))
print("bs%d: step before the process group exists %.3f ms, after %.3f ms" % (n, before, plain))
full = timed(lambda: eng.train_step(*batch, lr=3e-4, group=dist.group.WORLD))      # DP schedule, collectives executed on the communication stream
eng.grad_transport = "bf16"
half = timed(lambda: eng.train_step(*batch, lr=3e-4, group=dist.group.WORLD))      # ... with bf16 transport (pack + bf16 all-reduce + unpack)
print(("[three executor streams] " if three else "") + "bs%d: single-GPU step %.3f ms | DP schedule with one-rank RCCL all-reduces: f32 %.3f ms, bf16 transport %.3f ms" % (n, plain, full, half))
dist.destroy_process_group()
