"""world_size-2 gloo tests (CPU) of the data-parallel host path: mean all-reduce of the flat gradient
arenas, overlap handle semantics, and the equality '2 ranks x micro-batch == one process running both
micro-batches and averaging the gradients' (batch-norm statistics stay per micro-batch: SURVEY.md 7.7)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def fake_grads(batch, n):
  """Stand-in for one replica's backward: any deterministic function of the local batch."""
  b = torch.as_tensor(batch, dtype=torch.float32)
  w = torch.linspace(-1, 1, n)
  return torch.tanh(b.mean() * w) + b.std() * w ** 2


def _worker(rank, world, port, n, out):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from voicepuppet_amd.parallel import allreduce_mean, shard_batch
  rng = np.random.default_rng(0)
  global_batch = rng.normal(size=(8, 5)).astype(np.float32)
  lo, hi = shard_batch(8, rank, world)
  gd = fake_grads(global_batch[lo:hi], n)
  gg = fake_grads(global_batch[lo:hi] * 2, 3 * n)
  wd = allreduce_mean(gd, dist.group.WORLD, async_op=True)      # started before the "generator backward" ...
  # ... whose gradient arena travels as three contiguous buckets, last range first (engine.grad_buckets_g / train_step)
  a, b = 2 * n, n // 2
  works = [wd] + [allreduce_mean(gg[lo_:hi_], dist.group.WORLD, async_op=True) for lo_, hi_ in ((a, 3 * n), (b, a), (0, b))]
  for w in works:
    w.wait()                                                     # all complete before Adam
  if rank == 0:
    torch.save({"gd": gd, "gg": gg, "batch": torch.from_numpy(global_batch)}, out)
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_gradient_mean_equals_sequential_microbatches(tmp_path):
  n, world = 1000, 2
  out = str(tmp_path / "r0.pt")
  mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
  r = torch.load(out)
  r["batch"] = r["batch"].numpy()
  from voicepuppet_amd.parallel import shard_batch
  seq_d = sum(fake_grads(r["batch"][slice(*shard_batch(8, k, world))], n) for k in range(world)) / world
  seq_g = sum(fake_grads(r["batch"][slice(*shard_batch(8, k, world))] * 2, 3 * n) for k in range(world)) / world
  torch.testing.assert_close(r["gd"], seq_d, rtol=1e-6, atol=1e-7)
  torch.testing.assert_close(r["gg"], seq_g, rtol=1e-6, atol=1e-7)


def test_single_process_is_a_noop_and_shards_validate():
  from voicepuppet_amd.parallel import allreduce_mean, shard_batch
  t = torch.arange(4.0)
  allreduce_mean(t, None).wait()
  assert torch.equal(t, torch.arange(4.0))
  assert shard_batch(32, 3, 8) == (12, 16)
  with pytest.raises(ValueError):
    shard_batch(30, 0, 8)
