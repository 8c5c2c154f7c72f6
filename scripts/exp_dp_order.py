"""Does it matter whether the RCCL process group (its communicator and streams) exists BEFORE the step executor's streams are created?
One-rank RCCL group, data-parallel step with bf16 transport at BATCH frames, one variant per process:
  pg_first        init_process_group(device_id=...) - eager communicator - then the engine           (bench.py --gpus N until round 6)
  pg_first_lazy   init_process_group() without device_id - the communicator is created by the first collective, after the engine exists
  engine_first    the engine, then init_process_group(device_id=...)                                  (scripts/exp_dp1.py)
  warm_first      the engine AND ten plain steps (every executor stream has run), then the group
  reserved_first  vp_reserve_streams(), then init_process_group(device_id=...), then the engine   (parallel.init_distributed since round 6)
python scripts/exp_dp_order.py VARIANT [BATCH]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29521")
import torch
import torch.distributed as dist
import bench
from voicepuppet_amd.engine import PixReferEngine

variant = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def group(eager):
  kw = {"device_id": dev} if eager else {}
  dist.init_process_group("nccl", rank=0, world_size=1, **kw)


def engine():
  e = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  e.load_params(e.random_params(seed=0))
  e.grad_transport = "bf16"
  return e


batch = None
if variant == "pg_first":
  group(True); eng = engine()
elif variant == "pg_first_lazy":
  group(False); eng = engine()
elif variant == "engine_first":
  eng = engine(); group(True)
elif variant == "reserved_first":
  from voicepuppet_amd import _lib
  _lib.check(_lib.lib().vp_reserve_streams())
  group(True); eng = engine()
elif variant == "warm_first":
  eng = engine()
  batch = bench.synth_batch(n, 256, 1, dev)
  for _ in range(10): eng.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  group(True)
else:
  raise SystemExit("unknown variant " + variant)
if batch is None:
  batch = bench.synth_batch(n, 256, 1, dev)


def timed(fn, steps=40, warm=10):
  for _ in range(warm): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps * 1e3


plain = timed(lambda: eng.train_step(*batch, lr=3e-4))
dp = timed(lambda: eng.train_step(*batch, lr=3e-4, group=dist.group.WORLD))
print("%-14s batch %d: plain step %.3f ms, data-parallel step (one-rank RCCL, bf16 transport) %.3f ms" % (variant, n, plain, dp), flush=True)
eng.close()
dist.destroy_process_group()
