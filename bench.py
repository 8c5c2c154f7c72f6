#!/usr/bin/env python
"""Headline benchmark: PixReferNet G+D training step, frames/s at 256x256, bf16, per-GPU batch 32
(BASELINE.json metric).  One process per GPU; `python bench.py --gpus N --steps K --warmup W`
(N > 1: launched by torch.distributed.run, RCCL all-reduce of the G/D gradient arenas).

A step = forward (G, composite, 3xD, VGG trunk on 2N, all losses) + both backward passes + all-reduce
+ Adam(D) + Adam(G) on synthetic inputs already resident in HBM.  Weak scaling: the per-GPU batch is fixed.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel
(timed live with HIP events on the launch stream) and `cpu_baseline` (the numpy oracle, float32, timed on
the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

BF16_MFMA_PEAK = 2500.0   # TFLOP/s dense (MI355X_MICROARCH.md)
F32_MFMA_PEAK = 157.3
HBM_PEAK = 8000.0         # GB/s

# bench kernel name -> substring of the rocprofv3 kernel name (for the offline PMC traffic numbers)
def pmc_traffic(name):
  """HBM bytes per launch of the dominant kernel class from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
  separate runs of this same command, folded by scripts/pmc_summary.py; FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM');
  launch-weighted over the template variants of the class; None when not collected."""
  path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
  if not os.path.exists(path):
    return None
  rows = [r for r in json.load(open(path)) if r.get("class") == name]
  n = sum(r["launches"] for r in rows)
  return sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n if n else None


def synth_batch(n, h, seed, device):
  """U[0,1) box-blurred 5x5; soft-disc matte; fg = targets * masks packing."""
  g = torch.Generator(device="cpu").manual_seed(seed)

  def img(c):
    x = torch.rand(n, c, h, h, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="replicate"), 5, 1)
    return x.permute(0, 2, 3, 1).contiguous()
  yy, xx = torch.meshgrid(torch.arange(h), torch.arange(h), indexing="ij")
  r = ((yy - h / 2) ** 2 + (xx - h / 2) ** 2).float().sqrt()
  disc = ((0.35 * h + 4 - r) / 8).clamp(0, 1)
  masks = disc[None, :, :, None].expand(n, h, h, 3).contiguous()
  inputs, targets, ex_t = img(6), img(3), img(3)
  fg = torch.cat([ex_t * masks, targets * masks], dim=-1)
  return [t.to(device) for t in (inputs, fg, targets, masks)]


def cpu_baseline(h, n, ngf, ndf, steps=2):
  """The CPU restatement (oracle, float32, BLAS threads = all host cores) of the same G+D step: a bounded sample of `steps`
  steps at batch `n` (the reference trains at batch 2, train_pixrefer.py:36)."""
  from oracle import pixrefer_ref as ref
  p = ref.init_params(ngf, ndf, seed=0, dtype=np.float32)
  rng = np.random.default_rng(0)
  batch = [rng.uniform(size=(n, h, h, c)).astype(np.float32) for c in (6, 6, 3, 3)]
  st = ref.TrainState(p, ngf, ndf)
  t0 = time.time()
  for _ in range(steps):
    st.step(*batch)
  dt = time.time() - t0
  return {"value": n * steps / dt, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
          "sample": "%d G+D steps (fwd+bwd+Adam) of the numpy float32 oracle at batch %d, %dx%d, ngf=ndf=%d: %.1f s" % (steps, n, h, h, ngf, dt)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
  ap.add_argument("--height", type=int, default=256)
  ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-profile", action="store_true")
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  torch.cuda.set_device(local)
  device = torch.device("cuda", local)
  group = None
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=device)
    group = dist.group.WORLD

  from voicepuppet_amd.engine import PixReferEngine
  eng = PixReferEngine(args.batch, args.height, 64, 64, dtype=args.dtype, training=True)
  eng.load_params(eng.random_params(seed=0))   # the reference's initialisers, identical on every rank
  batch = synth_batch(args.batch, args.height, 1000 + rank, device)
  lr = 3e-4

  def sync():
    if world > 1:
      dist.barrier(group=group)
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    eng.train_step(*batch, lr=lr, group=group)
  sync()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    eng.train_step(*batch, lr=lr, group=group)
  sync()
  dt = time.perf_counter() - t0
  tmax = torch.tensor([dt], device=device)
  if world > 1:
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
  dt = float(tmax.item())
  ms = dt / args.steps * 1e3
  fps = args.batch * world * args.steps / dt

  roofline = None
  kernels = None
  if rank == 0 and not args.no_profile:
    # per-kernel launch durations, HIP events on the launch stream, over a few extra steps (outside the timed region)
    eng.profile(True)
    psteps = 3
    for _ in range(psteps):
      eng.train_step(*batch, lr=lr, group=None)
    torch.cuda.synchronize()
    recs = eng.profile_collect()
    eng.profile(False)
    recs.sort(key=lambda r: -r["ms"])
    kernels = [{"name": r["name"], "calls_per_step": r["calls"] / psteps, "ms_per_step": r["ms"] / psteps,
                "tflops": r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0} for r in recs]
    top = recs[0]
    peak = BF16_MFMA_PEAK if args.dtype == "bf16" else F32_MFMA_PEAK
    ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
    roofline = {"kernel": top["name"], "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": pmc_traffic(top["name"]), "avg_launch_ms": top["ms"] / top["calls"],
                "launches_per_step": top["calls"] / psteps, "algorithmic_bytes_per_launch": top["bytes"] / top["calls"],
                "algorithmic_flops_per_launch": top["flops"] / top["calls"]}
  if world > 1:
    dist.barrier(group=group)

  if rank == 0:
    out = {"metric": "PixReferNet G+D step frames/sec @256x256 bs=32", "value": fps, "unit": "frames/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
           "data": "synthetic",
           "config": {"workload": "PixReferNet G+D training step (G + 3xD + VGG16-conv3_3 perceptual + TF-Adam x2), "
                                  "%dx%d, ngf=ndf=64, per-GPU batch %d" % (args.height, args.height, args.batch),
                      "global_batch": args.batch * world, "parallelism": "dp%d" % world},
           "roofline": roofline, "kernels": kernels,
           "step_tflops": 163.02e9 * (args.height / 256) ** 2 * args.batch * world / (ms * 1e-3) / 1e12}
    if not args.no_cpu_baseline and world == 1:
      out["cpu_baseline"] = cpu_baseline(args.height, 2, 64, 64)
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
