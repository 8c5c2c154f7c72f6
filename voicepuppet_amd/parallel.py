"""Data-parallel plumbing: the gradient exchange of the G+D step over torch.distributed.

One process per GPU; backend "nccl" is RCCL over xGMI on MI355X.  The step's only collective is the
mean all-reduce of the two flat f32 gradient arenas (discriminator 2.77 M floats, generator 35.16 M
floats); batch-norm statistics stay per replica.  RCCL reduces with ncclAvg, so no extra
scaling kernel runs on the device; the gloo branch (CPU tests) sums and scales.
"""
import torch
import torch.distributed as dist


class _Done(object):
  def wait(self):
    return True


class _SumThenScale(object):
  def __init__(self, work, t, world):
    self.work, self.t, self.world = work, t, world

  def wait(self):
    self.work.wait()
    self.t.div_(self.world)
    return True


def allreduce_mean(t, group=None, async_op=False, skip_single=True):
  """In-place mean of tensor `t` over the group; returns an object with .wait().  A one-rank group is a no-op unless
  skip_single=False (tests drive the real collective path on one GPU that way)."""
  if group is None and not dist.is_initialized():
    return _Done()
  world = dist.get_world_size(group)
  if world == 1 and skip_single:
    return _Done()
  if dist.get_backend(group) == "nccl":
    w = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=True)
    if not async_op:
      w.wait()
      return _Done()
    return w
  w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
  h = _SumThenScale(w, t, world)
  if not async_op:
    h.wait()
    return _Done()
  return h


def shard_batch(global_batch, rank, world):
  """Samples [lo, hi) of a global batch owned by `rank` (even split; weak scaling keeps hi-lo fixed)."""
  if global_batch % world:
    raise ValueError("global batch %d is not divisible by %d ranks" % (global_batch, world))
  per = global_batch // world
  return rank * per, (rank + 1) * per


def shard_round_robin(n_items, rank, world):
  """Indices of `n_items` independent work items (clips) owned by `rank`: rank, rank + world, ...  No collective is involved:
  the items are independent (log-mel / BFMNet / infer_bfmvid shard by clip, replicas only)."""
  if not 0 <= rank < world:
    raise ValueError("rank %d outside a world of %d" % (rank, world))
  return list(range(rank, n_items, world))
