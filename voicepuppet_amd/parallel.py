"""Data-parallel plumbing: the gradient exchange of the G+D step over torch.distributed.

One process per GPU; backend "nccl" is RCCL over xGMI on MI355X.  The step's only collective is the
mean all-reduce of the two flat f32 gradient arenas (discriminator 2.77 M floats, generator 35.16 M
floats); batch-norm statistics stay per replica.  RCCL reduces with ncclAvg, so no extra
scaling kernel runs on the device; the gloo branch (CPU tests) sums and scales.
"""
import collections
import datetime
import logging
import os
import sys
import threading
import time

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)

WATCHDOG_EXIT_CODE = 75        # EX_TEMPFAIL: "try again" - the restart is a FRESH process from the last checkpoint


def init_distributed(backend="nccl", timeout_s=None, **kw):
  """dist.init_process_group for one process per GPU with the failure behaviour SURVEY.md 5 asks for: a collective that errors or
  does not complete within `timeout_s` (default VP_COLLECTIVE_TIMEOUT_S or 600) aborts the communicator and raises in this rank
  (TORCH_NCCL_ASYNC_ERROR_HANDLING=1: torch's RCCL watchdog thread tears the process down), the launcher (torch.distributed.run)
  then stops the other ranks, and every rank's exit status is non-zero.  Nothing is re-executed in place: a process that has
  touched the GPU is never exec'd over; the job is started again from the last checkpoint (train_pixrefer.py --resume)."""
  if timeout_s is None:
    timeout_s = float(os.environ.get("VP_COLLECTIVE_TIMEOUT_S", "600"))
  os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  if backend == "nccl":
    # the step executor's streams FIRST (the caller has selected its device): streams created behind the communicator's get the HIP
    # runtime's leftover hardware queues and every step of the job runs 8 % (32 frames per GPU) to 30 % (4 frames) slow
    # (include/vp_hip.h vp_reserve_streams, scripts/exp_dp_order.py)
    from . import _lib
    _lib.check(_lib.lib().vp_reserve_streams(), "vp_reserve_streams")
  dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
  return dist.group.WORLD


class StepWatchdog(object):
  """Rank-liveness watchdog of a training loop ('RCCL error / timeout -> abort all ranks; restart from the last
  checkpoint').  The step never blocks the host, so a peer that died inside a collective shows up as a device stream that stops
  making progress (RCCL kernels spin) while the host keeps enqueuing, or - with a host-blocking backend (gloo) - as a host that
  stops calling beat().  Both are watched from a daemon thread:

    beat(event)  once per step, `event` = a torch.cuda.Event recorded behind the step (None: the step is complete on return);
    the thread fires when the OLDEST unfinished step was enqueued more than `timeout_s` ago, or when beat() has not been called
    for `timeout_s` (after the first beat).

  Firing logs the rank, the last finished step and the age, then ends the process with WATCHDOG_EXIT_CODE through os._exit (no
  destructors: the communicator may be wedged); torch.distributed.run sees the non-zero status and stops the remaining ranks.
  timeout_s <= 0 disables it.  on_timeout(info) replaces the exit (tests)."""

  def __init__(self, timeout_s=None, rank=0, on_timeout=None, poll_s=0.25, device=None):
    # device: the CUDA device index whose events the thread queries (an Event.query() from a thread that never selected a device
    # would otherwise run against device 0's context on every rank)
    self.device = device
    self.paused = 0
    if timeout_s is None:
      timeout_s = float(os.environ.get("VP_WATCHDOG_TIMEOUT_S", "300"))
    self.timeout_s, self.rank, self.on_timeout, self.poll_s = float(timeout_s), rank, on_timeout, poll_s
    self.pending = collections.deque()
    self.lock = threading.Lock()
    self.steps_enqueued = self.steps_done = 0
    self.last_beat = None
    self.fired = None
    self._stop = threading.Event()
    self.thread = None
    if self.timeout_s > 0:
      self.thread = threading.Thread(target=self._run, name="vp-step-watchdog", daemon=True)
      self.thread.start()

  def beat(self, event=None):
    now = time.monotonic()
    with self.lock:
      self.steps_enqueued += 1
      self.last_beat = now
      self.pending.append((now, event, self.steps_enqueued))

  def pause(self):
    """A long host-side section (checkpoint save, first-step initialisation, an evaluation pass) is not a hung rank: the 'host' condition
    is suspended until resume(); steps already enqueued keep being watched on the device side."""
    with self.lock:
      self.paused += 1

  def resume(self):
    with self.lock:
      self.paused = max(0, self.paused - 1)
      if self.last_beat is not None:
        self.last_beat = time.monotonic()

  def _drain(self):
    with self.lock:
      while self.pending:
        t, ev, k = self.pending[0]
        if ev is not None and not ev.query():
          break
        self.pending.popleft()
        self.steps_done = k

  def _check(self, now):
    self._drain()
    with self.lock:
      if self.pending and now - self.pending[0][0] > self.timeout_s:
        return {"why": "device", "rank": self.rank, "age_s": now - self.pending[0][0], "last_finished_step": self.steps_done,
                "oldest_unfinished_step": self.pending[0][2]}
      if not self.paused and self.last_beat is not None and now - self.last_beat > self.timeout_s:
        return {"why": "host", "rank": self.rank, "age_s": now - self.last_beat, "last_finished_step": self.steps_done,
                "oldest_unfinished_step": self.steps_done + 1}
    return None

  def _run(self):
    if self.device is not None:
      try:
        import torch
        torch.cuda.set_device(self.device)
      except Exception:          # (CPU-only tests drive the watchdog with event = None)
        pass
    while not self._stop.wait(self.poll_s):
      info = self._check(time.monotonic())
      if info is None:
        continue
      self.fired = info
      logger.error("rank %d: no progress for %.1f s (%s stalled: step %d has not finished, last finished step %d) - a peer is gone or a "
                   "collective hangs; ending this rank with status %d, restart the job from the last checkpoint",
                   info["rank"], info["age_s"], info["why"], info["oldest_unfinished_step"], info["last_finished_step"], WATCHDOG_EXIT_CODE)
      if self.on_timeout is not None:
        self.on_timeout(info)
        return
      for h in logging.getLogger().handlers:
        h.flush()
      sys.stderr.flush()
      os._exit(WATCHDOG_EXIT_CODE)

  def close(self):
    self._stop.set()
    if self.thread is not None:
      self.thread.join(timeout=2.0)


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
    self.timing = False           # True: HIP events on the communication stream around every bucket (bucket_ms)
    self.marks = []

  def _buffer(self, t):
    key = (t.data_ptr(), t.numel())
    b = self.buffers.get(key)
    if b is None:
      b = self.buffers[key] = torch.empty(t.numel(), dtype=torch.bfloat16, device=t.device)
    return b

  def begin_step(self):
    self.marks = []

  def bucket_ms(self):
    """After a synchronize, with timing on: per bucket of the last step, in issue order, {name, bytes (on the wire per rank and
    direction), wait_ms (the communication stream idle until the bucket's gradients were final, measured from the previous bucket's
    end), allreduce_ms (pack + collective + unpack), update_ms (the Adam update + re-pack behind it)} - HIP events on the
    communication stream, so a scaling record can say where the exchange's time goes."""
    out, prev = [], None
    for name, nbytes, e0, e1, e2 in self.marks:
      out.append({"name": name, "bytes": nbytes, "wait_ms": round(prev.elapsed_time(e0), 4) if prev is not None else None,
                  "allreduce_ms": round(e0.elapsed_time(e1), 4), "update_ms": round(e1.elapsed_time(e2), 4)})
      prev = e2
    return out

  def start(self, t, then=None, name=""):
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
      if self.timing:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record(self.stream)
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
      if self.timing:
        ev[1].record(self.stream)
      if then is not None:
        then(sp)
      if self.timing:
        ev[2].record(self.stream)
        self.marks.append((name, t.numel() * (2 if self.transport == "bf16" else 4), ev[0], ev[1], ev[2]))

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
