"""Where the data-parallel schedule's extra time goes, on ONE GPU with a one-rank RCCL group: per bucket of a step, in issue order, the
time the communication (= side) stream waited for the bucket, its all-reduce (pack + collective + unpack) and its Adam update + re-pack
(HIP events on that stream, parallel.GradExchange.bucket_ms).   python scripts/exp_dp_buckets.py [batch]   (one batch per process: a second
engine in the same process measures slow - EXPERIMENTS.md 0.8 of round 6)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
import torch
import torch.distributed as dist
import bench
from voicepuppet_amd.engine import PixReferEngine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for n in [int(a) for a in sys.argv[1:2]] or [4]:
  eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  eng.load_params(eng.random_params(seed=0))
  eng.grad_transport = "bf16"
  batch = bench.synth_batch(n, 256, 1, dev)
  for _ in range(10): eng.train_step(*batch, lr=3e-4, group=dist.group.WORLD)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(40): eng.train_step(*batch, lr=3e-4, group=dist.group.WORLD)
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / 40 * 1e3
  eng._exchange.timing = True
  for _ in range(2): eng.train_step(*batch, lr=3e-4, group=dist.group.WORLD)
  torch.cuda.synchronize()
  print("batch %d: data-parallel step %.3f ms; buckets:" % (n, ms))
  for b in eng._exchange.bucket_ms(): print("   ", b)
  eng.close()
dist.destroy_process_group()
