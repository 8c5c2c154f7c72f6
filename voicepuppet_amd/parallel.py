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


class GradExchange(object):
  """The gradient exchange of one training step: buckets of the flat f32 gradient arenas are all-reduced (mean) while the backward
  pass still computes the next ones.

  Ordering is explicit (include/vp_hip.h, 'STREAM-ORDER CONTRACT'): start() records an event on the CURRENT stream - the one the
  executor was called on, which by then waits for every kernel that wrote the bucket - and the collective is issued on a
  communication stream that waits for that event; finish() makes the current stream wait for the communication stream, so the
  Adam kernels that follow see the reduced gradients.  Nothing here blocks the host with the RCCL backend.

  transport = 'f32': RCCL averages the arena range in place (ncclAvg).  transport = 'bf16': the range is rounded to bf16 into a
  communication buffer (vp_grad_pack_bf16), summed as bf16, and written back as f32 times 1 / world (vp_grad_unpack_bf16): half
  the bytes on xGMI, one bf16 rounding per element and rank; the f32 arena, Adam state and parameters stay f32."""

  def __init__(self, group, transport="f32", stream=None):
    """stream: a torch stream to issue the collectives on (the executor's side stream, engine.py); default: a stream of its own"""
    if transport not in ("f32", "bf16"):
      raise ValueError("gradient transport must be 'f32' or 'bf16', got %r" % (transport,))
    self.group, self.transport = group, transport
    self.world = dist.get_world_size(group)
    self.backend = dist.get_backend(group)
    self.stream = stream if stream is not None else torch.cuda.Stream()
    self.buffers = {}

  def _buffer(self, t):
    key = (t.data_ptr(), t.numel())
    b = self.buffers.get(key)
    if b is None:
      b = self.buffers[key] = torch.empty(t.numel(), dtype=torch.bfloat16, device=t.device)
    return b

  def start(self, t, then=None):
    """Mean all-reduce of the contiguous f32 range `t` on the communication stream, ordered behind everything enqueued on the
    current stream; `then(stream_pointer)` is called right behind it to enqueue work that consumes the reduced range on that
    stream (the bucket's Adam update)."""
    import ctypes
    from . import _lib
    ready = torch.cuda.Event()
    ready.record()                                   # on the stream the executor was driven from
    with torch.cuda.stream(self.stream):
      self.stream.wait_event(ready)
      sp = ctypes.c_void_p(self.stream.cuda_stream)
      # (a blocking-style call: with RCCL it only orders the communication stream behind the collective, the host does not wait)
      if self.transport == "bf16":
        buf = self._buffer(t)
        _lib.check(_lib.lib().vp_grad_pack_bf16(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(buf.data_ptr()), t.numel(), sp), "vp_grad_pack_bf16")
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        _lib.check(_lib.lib().vp_grad_unpack_bf16(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(t.data_ptr()), t.numel(), 1.0 / self.world, sp),
                   "vp_grad_unpack_bf16")
      elif self.backend == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
      else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.div_(self.world)
      if then is not None:
        then(sp)

  def finish(self):
    """Everything started so far is ordered before what the current stream does next."""
    torch.cuda.current_stream().wait_stream(self.stream)


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
