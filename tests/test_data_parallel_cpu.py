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


# ---- rank liveness ("RCCL error / timeout -> abort all ranks; restart from the last checkpoint") ----------------
class _FakeEvent(object):
  def __init__(self):
    self.done = False

  def query(self):
    return self.done


def test_watchdog_fires_on_a_stalled_device_step_and_on_a_silent_host():
  import time
  from voicepuppet_amd.parallel import StepWatchdog
  fired = []
  dog = StepWatchdog(timeout_s=0.6, rank=3, on_timeout=fired.append, poll_s=0.05)
  evs = [_FakeEvent() for _ in range(4)]
  for e in evs[:3]:
    dog.beat(e)
  evs[0].done = evs[1].done = True             # steps 1, 2 finish; step 3 never does (a peer died inside its collective)
  t0 = time.monotonic()
  while not fired and time.monotonic() - t0 < 5:
    time.sleep(0.05)
    dog.beat(_FakeEvent())                      # the host keeps enqueuing: only the device is stuck
  dog.close()
  assert fired and fired[0]["why"] == "device" and fired[0]["rank"] == 3
  assert fired[0]["last_finished_step"] == 2 and fired[0]["oldest_unfinished_step"] == 3
  # host side: beats stop (a host-blocking collective that never returns)
  fired2 = []
  dog2 = StepWatchdog(timeout_s=0.4, rank=0, on_timeout=fired2.append, poll_s=0.05)
  dog2.beat(None)
  time.sleep(1.2)
  dog2.close()
  assert fired2 and fired2[0]["why"] == "host" and fired2[0]["last_finished_step"] == 1
  # healthy loop: never fires; timeout 0 disables the thread altogether
  fired3 = []
  dog3 = StepWatchdog(timeout_s=0.5, on_timeout=fired3.append, poll_s=0.05)
  for _ in range(20):
    dog3.beat(None)
    time.sleep(0.05)
  dog3.close()
  assert not fired3
  assert StepWatchdog(timeout_s=0).thread is None


def _liveness_worker(rank, world, port):
  import time
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  from voicepuppet_amd.parallel import StepWatchdog, init_distributed, allreduce_mean
  group = init_distributed("gloo", timeout_s=60, rank=rank, world_size=world)
  dog = StepWatchdog(timeout_s=1.5, rank=rank, poll_s=0.1)          # default action: os._exit(WATCHDOG_EXIT_CODE)
  g = torch.ones(16)
  for step in range(1000):
    if rank == 1 and step == 3:
      time.sleep(3600)                                               # the peer hangs (it is killed by the parent below)
    allreduce_mean(g, group)                                         # gloo blocks the host here once the peer is gone
    dog.beat(None)
  os._exit(0)                                                        # not reached by rank 0


def test_a_hung_peer_ends_the_surviving_rank_with_the_watchdog_status():
  from voicepuppet_amd.parallel import WATCHDOG_EXIT_CODE
  ctx = mp.get_context("spawn")
  port = _free_port()
  procs = [ctx.Process(target=_liveness_worker, args=(r, 2, port)) for r in range(2)]
  for p in procs:
    p.start()
  procs[0].join(timeout=60)
  try:
    assert procs[0].exitcode == WATCHDOG_EXIT_CODE, procs[0].exitcode   # non-zero, from the watchdog - not a hang, not a clean exit
  finally:
    for p in procs:
      if p.is_alive():
        p.kill()                                                        # exact process objects we started
      p.join(timeout=10)
