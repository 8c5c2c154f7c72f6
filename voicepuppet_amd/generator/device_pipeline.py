"""PixReferNet input pipeline on the device (reference: generator/generator.py:956-1019).

The reference prepares every training sample on the host: two cv2.imread's, BGR->RGB, a random square crop of the S x 3S
triptych, cv2.resize back to S x S and the 6-channel packing - float32, 18 * S * S values per sample, then a host-to-device
copy of all of it.  At thousands of frames per second that thread is the bottleneck.  Here the host only hands over the decoded
uint8 frames (2 * 9 * S * S bytes per sample: 8x fewer bytes over PCIe) and three integers per crop; libvp_hip.so's
vp_pixrefer_pack_frames does the rest in one kernel, and `FramePrefetcher` overlaps the copies of batch k+1 with step k on a
second HIP stream (pinned staging buffers, event hand-off).
"""
import ctypes
import random

import numpy as np
import torch

from .. import _lib


def draw_crop(img_size, crop_ratio, rng=random):
  """(rx, ry, rsize) exactly as generator.py:975-977 / 994-996 draws them."""
  rsize = rng.randint(int(img_size * crop_ratio), img_size)
  rx = rng.randint(0, img_size - rsize)
  ry = rng.randint(0, img_size - rsize)
  return rx, ry, rsize


class DeviceFramePacker:
  """uint8 triptych frames + crops -> (inputs, fg_inputs, targets, masks) float32 device tensors."""

  def __init__(self, batch, img_size, device=None):
    if not torch.cuda.is_available():
      raise RuntimeError("DeviceFramePacker needs an MI355X (no CPU fallback)")
    self.L = _lib.lib()
    self.batch, self.S = batch, img_size
    self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    f = lambda c: torch.empty(batch, img_size, img_size, c, dtype=torch.float32, device=self.device)
    self.out = (f(6), f(6), f(3), f(3))

  def __call__(self, ex_u8, cur_u8, crops):
    """ex_u8 / cur_u8: [N, S, 3S, 3] uint8 device tensors (BGR, as cv2.imread); crops: [N, 2, 3] int32 device tensor."""
    N, S = self.batch, self.S
    assert ex_u8.shape == (N, S, 3 * S, 3) and cur_u8.shape == ex_u8.shape and ex_u8.dtype == torch.uint8 and cur_u8.dtype == torch.uint8
    assert crops.shape == (N, 2, 3) and crops.dtype == torch.int32
    ex_u8, cur_u8, crops = ex_u8.contiguous(), cur_u8.contiguous(), crops.contiguous()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(self.L.vp_pixrefer_pack_frames(p(ex_u8), p(cur_u8), p(crops), N, S, p(self.out[0]), p(self.out[1]), p(self.out[2]),
                                              p(self.out[3]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "vp_pixrefer_pack_frames")
    return self.out


class FramePrefetcher:
  """Double-buffered host -> device hand-over of uint8 frame batches on a side stream.

  `source` yields (ex [N,S,3S,3] uint8, cur [N,S,3S,3] uint8, crops [N,2,3] int32) as numpy arrays or torch tensors; pinned
  torch tensors cross PCIe straight from where they are (they must stay unchanged until two batches later).  next() returns the four packed
  float32 tensors of the oldest batch in flight (valid until the next call) and starts the copies of a following one; the
  copies and the pack kernel of batch k+1 overlap the training step of batch k.
  """

  def __init__(self, source, batch, img_size, depth=2, device=None):
    self.source = iter(source)
    self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    # the process's fourth executor stream (include/vp_hip.h vp_host_stream): the training engine beside this prefetcher keeps to three
    # (vp_pixrefer_use_streams), a stream created here would be a fifth one on a shared hardware queue
    sp = _lib.lib().vp_host_stream()
    if not sp:
      _lib.check(-1, "vp_host_stream")
    self.stream = torch.cuda.ExternalStream(sp, device=self.device)
    S = img_size
    self.slots = []
    for _ in range(depth):
      host = (torch.empty(batch, S, 3 * S, 3, dtype=torch.uint8).pin_memory(), torch.empty(batch, S, 3 * S, 3, dtype=torch.uint8).pin_memory(),
              torch.empty(batch, 2, 3, dtype=torch.int32).pin_memory())
      dev = tuple(torch.empty_like(h, device=self.device) for h in host)
      self.slots.append({"host": host, "dev": dev, "packer": DeviceFramePacker(batch, img_size, self.device),
                         "ready": torch.cuda.Event(), "free": torch.cuda.Event(), "busy": False})
    self.head = self.tail = 0
    self._copied = []                                 # copy-done events of the last batches whose sources were read in place
    for _ in range(depth):
      self._fill()

  def _fill(self):
    slot = self.slots[self.head % len(self.slots)]
    # The contract for pinned sources read in place: a buffer stays unchanged "until two batches later".  Enqueuing is not
    # executing: nothing else throttles a host that runs ahead of the device (a launcher that reads no loss per step), so before the
    # source may produce batch k - possibly into the buffer batch k-3 or older was read from - the H2D copies of batch k-2 (and, in
    # stream order, of everything before it) must have EXECUTED.  They normally have; the wait is then a query.
    if len(self._copied) >= 2:
      self._copied[-2].synchronize()
    try:
      ex, cur, crops = next(self.source)
    except StopIteration:
      return False
    # a source that already decodes into PINNED torch tensors (what a decoder thread should do) is copied from directly; anything
    # else is staged through this slot's pinned buffers first (a host memcpy of 18 * S * S bytes per sample on the calling thread)
    srcs = []
    staged = False
    for h, a in zip(slot["host"], (ex, cur, crops)):
      if isinstance(a, torch.Tensor) and a.is_pinned() and a.dtype == h.dtype and a.shape == h.shape and a.is_contiguous():
        srcs.append(a)
        continue
      if not staged and slot["busy"]:
        slot["ready"].synchronize()                   # the previous copies out of this slot's pinned buffers have completed
      staged = True
      h.copy_(a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)))
      srcs.append(h)
    slot["src"] = srcs                                # keeps a caller's pinned tensors alive until the slot is refilled
    with torch.cuda.stream(self.stream):
      if slot["busy"]:
        self.stream.wait_event(slot["free"])          # the consumer of this slot's previous batch has been enqueued past it
      for d, h in zip(slot["dev"], srcs):
        d.copy_(h, non_blocking=True)
      if not staged:
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self._copied = self._copied[-2:] + [ev]
      slot["packer"](*slot["dev"])
      slot["ready"].record(self.stream)
    slot["busy"] = True
    self.head += 1
    return True

  def next(self):
    if self.tail == self.head:
      raise StopIteration
    # the slot handed out by the previous call is free once everything enqueued so far on the compute stream has run
    if self.tail > 0:
      prev = self.slots[(self.tail - 1) % len(self.slots)]
      prev["free"].record(torch.cuda.current_stream())
      self._fill()
    slot = self.slots[self.tail % len(self.slots)]
    torch.cuda.current_stream().wait_event(slot["ready"])
    self.tail += 1
    return slot["packer"].out

  __next__ = next

  def __iter__(self):
    return self

