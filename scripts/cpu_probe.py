"""How many host cores does the GPU box really give this process, and how does the torch-CPU proxy scale with threads?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import pixrefer_ref as ref
from oracle.pixrefer_torch import TorchGraph
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", bench.usable_cores())
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e: print("cpu.max", e)
print("loadavg", open("/proc/loadavg").read().strip())
p = ref.init_params(64, 64, seed=0, dtype=np.float32)
rng = np.random.default_rng(0)
b2 = [rng.uniform(size=(2, 256, 256, c)).astype(np.float32) for c in (6, 6, 3, 3)]
for th in (8, 16, 32, 64):
  torch.set_num_threads(th)
  tg = TorchGraph(p, 64, 64, torch.float32)
  tg.step(*b2)
  t = time.time(); tg.step(*b2); dt = time.time() - t
  print("torch threads %d: bs2 step %.2f s" % (th, dt), flush=True)
  if dt > 20: break
