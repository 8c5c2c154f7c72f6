"""Host-side driver of the PixReferNet step executor (libvp_hip.so: vp_pixrefer_*).

torch is used for device memory, the current HIP stream and (optionally) torch.distributed; all
arithmetic happens in the HIP kernels behind the C ABI.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import VP_BF16, VP_F32, PixReferDesc


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def manifest(desc, which):
  """[(tf_variable_name, offset, shape)] of arena `which` (0 generator, 1 discriminator, 2 vgg_16)."""
  L = _lib.lib()
  out = []
  name = ctypes.create_string_buffer(256)
  off = ctypes.c_size_t()
  nd = ctypes.c_int()
  shp = (ctypes.c_int64 * 4)()
  i = 0
  while L.vp_pixrefer_param_info(ctypes.byref(desc), which, i, name, 256, ctypes.byref(off), ctypes.byref(nd), shp) == 0:
    out.append((name.value.decode(), int(off.value), tuple(int(shp[k]) for k in range(nd.value))))
    i += 1
  return out


class PixReferEngine:
  """One replica of the PixReferNet graph (pixrefer.py:356-438) on the current device."""

  def __init__(self, batch, height, ngf=64, ndf=64, dtype="bf16", training=True, l1_weight=500.0, gan_weight=1.0,
               device=None, per_sample_bn=False, streams=0, d_backward_fork=0, d_beside_vgg=0):
    if not torch.cuda.is_available():
      raise RuntimeError("PixReferEngine needs an MI355X (no CPU fallback)")
    self.L = _lib.lib()
    self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    self.desc = PixReferDesc(batch, height, ngf, ndf, VP_BF16 if dtype == "bf16" else VP_F32, 1 if training else 0,
                             l1_weight, gan_weight, 1 if (per_sample_bn and not training) else 0,
                             int(streams), int(d_backward_fork), int(d_beside_vgg))   # schedule fields: 0 = default (include/vp_hip.h)
    self.training = training
    self.compute_dtype = torch.bfloat16 if dtype == "bf16" else torch.float32
    d = ctypes.byref(self.desc)
    self.manifests = [manifest(self.desc, w) for w in range(3)]
    counts = [self.L.vp_pixrefer_param_count(d, w) for w in range(3)]
    z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
    self.params_g = z(counts[0])
    self.params_d = z(counts[1]) if training else None
    self.params_vgg = z(counts[2]) if training else None
    self.grads_g = z(counts[0]) if training else None
    self.grads_d = z(counts[1]) if training else None
    if training:
      self.adam = {"g": [z(counts[0]), z(counts[0])], "d": [z(counts[1]), z(counts[1])]}
    self.t_g = self.t_d = 0
    self._dp_streams_set = False
    self.dp_own_stream = False           # data parallel: the collectives on a stream of the exchange's own instead of the executor's side stream (slower here: DESIGN.md 5)
    self.grad_transport = "f32"          # data parallel: 'bf16' halves the bytes of the gradient all-reduce (parallel.GradExchange)
    self._exchange = None
    self.fused_update = True             # single-GPU train_step: vp_pixrefer_backward_update (backward + Adam x 2 + re-pack in one call)
    ws = self.L.vp_pixrefer_workspace_bytes(d)
    if ws == 0:
      raise ValueError("invalid PixReferNet descriptor: %s" % self.L.vp_last_error().decode())
    self.workspace = torch.zeros(ws, dtype=torch.uint8, device=self.device)
    h = ctypes.c_void_p()
    _lib.check(self.L.vp_pixrefer_create(d, _ptr(self.workspace), ws, _ptr(self.params_g), _ptr(self.params_d),
                                         _ptr(self.params_vgg), _ptr(self.grads_g), _ptr(self.grads_d), _stream(),
                                         ctypes.byref(h)), "vp_pixrefer_create")
    self.h = h
    self._keep = None

  def close(self):
    """Destroys the plan NOW and returns its 7 GB workspace's owner to the collector.  (The executor's HIP streams are process-wide and
    stay: until they were, an engine created while an earlier one's streams existed ran slow - scripts/exp_engine_sequence.py,
    profiles/r06_exp_engine_sequence.txt.)"""
    if getattr(self, "h", None):
      self.L.vp_pixrefer_destroy(self.h)
      self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  # ---- parameters -------------------------------------------------------------------------------
  def arena(self, which):
    return [self.params_g, self.params_d, self.params_vgg][which]

  def fill_arena(self, dst, which, params, suffix=""):
    """Write {tf_variable_name + suffix: array} into the flat f32 tensor `dst` laid out like arena `which`; names that are
    absent keep their current value.  Returns the number of variables written."""
    host = dst.cpu().numpy()
    hit = 0
    for name, off, shape in self.manifests[which]:
      key = name + suffix
      if key in params:
        v = np.asarray(params[key], dtype=np.float32)
        assert v.shape == shape, (key, v.shape, shape)
        host[off:off + v.size] = v.reshape(-1)
        hit += 1
    dst.copy_(torch.from_numpy(host))
    return hit

  def load_params(self, params):
    """params: {tf_variable_name: numpy array}; missing names keep their current value."""
    for which in range(3):
      a = self.arena(which)
      if a is not None:
        self.fill_arena(a, which, params)
    self.params_changed()

  def load_adam(self, params, t_g=None, t_d=None):
    """Optimiser slots in tf.train.Saver naming: '<variable>/Adam' (m) and '<variable>/Adam_1' (v) (pixrefer.py:398,405);
    t_* = the number of updates already applied (TF keeps beta1_power / beta2_power = beta ** (t + 1) instead)."""
    for key, which in (("g", 0), ("d", 1)):
      for idx, tag in ((0, "/Adam"), (1, "/Adam_1")):
        self.fill_arena(self.adam[key][idx], which, params, tag)
    if t_g is not None:
      self.t_g = int(t_g)
    if t_d is not None:
      self.t_d = int(t_d)

  def random_params(self, seed=0):
    """The reference's variable initialisers (pixrefer.py:64,68,100-101: kernels N(0,0.02), gamma N(1,0.02), bias/beta 0)
    plus He-normal stand-ins for the external vgg_16 checkpoint; {tf_variable_name: float32 array}, same on every rank."""
    rng = np.random.default_rng(seed)
    p = {}
    for which in range(3):
      if self.arena(which) is None:
        continue
      for name, _, shape in self.manifests[which]:
        if name.endswith("kernel"):
          p[name] = rng.normal(0, 0.02, shape).astype(np.float32)
        elif name.endswith("gamma"):
          p[name] = rng.normal(1.0, 0.02, shape).astype(np.float32)
        elif name.endswith("weights"):
          p[name] = rng.normal(0, np.sqrt(2.0 / (shape[0] * shape[1] * shape[2])), shape).astype(np.float32)
        else:
          p[name] = np.zeros(shape, np.float32)
    return p

  def get_params(self, which, src=None):
    a = (self.arena(which) if src is None else src).cpu().numpy()
    return {name: a[off:off + int(np.prod(shape))].reshape(shape).copy() for name, off, shape in self.manifests[which]}

  def params_changed(self):
    _lib.check(self.L.vp_pixrefer_params_changed(self.h))

  # ---- execution --------------------------------------------------------------------------------
  def forward(self, inputs, fg_inputs, targets, masks=None):
    """float32 NHWC device tensors in [0,1] (generator.py:1011-1019 layout)."""
    N, H = self.desc.batch, self.desc.height
    assert inputs.shape == (N, H, H, 6) and targets.shape == (N, H, H, 3)
    ts = [t.contiguous() for t in (inputs, fg_inputs, targets)]
    m = masks.contiguous() if masks is not None else None
    for t in ts + ([m] if m is not None else []):
      assert t.dtype == torch.float32 and t.is_cuda
    self._keep = (ts, m)   # the backward reads targets/masks again
    if fg_inputs.shape[-1] == 3:   # infer_bfmvid.py:203 feeds a 3-channel foreground reference: the library reads it as it is
      assert fg_inputs.shape == (N, H, H, 3) and not self.training
      _lib.check(self.L.vp_pixrefer_forward_fg3(self.h, _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), _stream()), "vp_pixrefer_forward_fg3")
      return
    assert fg_inputs.shape == (N, H, H, 6)
    _lib.check(self.L.vp_pixrefer_forward(self.h, _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), _ptr(m), _stream()),
               "vp_pixrefer_forward")

  def backward(self):
    _lib.check(self.L.vp_pixrefer_backward(self.h, _stream()), "vp_pixrefer_backward")

  def backward_d(self):
    _lib.check(self.L.vp_pixrefer_backward_d(self.h, _stream()), "vp_pixrefer_backward_d")

  def backward_g(self):
    _lib.check(self.L.vp_pixrefer_backward_g(self.h, _stream()), "vp_pixrefer_backward_g")

  def backward_g_stage(self, stage):
    """Stage `stage` of backward_g (0 .. backward_g_stages()-1); see grad_buckets_g."""
    _lib.check(self.L.vp_pixrefer_backward_g_stage(self.h, int(stage), _stream()), "vp_pixrefer_backward_g_stage")

  def grad_buckets_g(self):
    """[(lo, hi)] float ranges of the generator gradient arena that are final after stage 0, 1, 2 of backward_g:
    the arena is in TF variable order (encoders, merged encoders, merged decoders, decoders) and the backward pass
    walks it from the end, so each stage completes a contiguous suffix."""
    off = {name: o for name, o, _ in self.manifests[0]}
    a = off["generator/merged_decoder_5/conv2d_transpose/kernel"]
    b = off["generator/merged_encoder_2/conv2d/kernel"]
    return [(a, self.grads_g.numel()), (b, a), (0, b)]

  def use_streams(self, n):
    """3 or 4 executor streams (include/vp_hip.h vp_pixrefer_use_streams): a caller that feeds the step through a prefetcher with a stream
    of its own (generator/device_pipeline.FramePrefetcher) asks for 3."""
    _lib.check(self.L.vp_pixrefer_use_streams(self.h, int(n)), "vp_pixrefer_use_streams")

  def train_step(self, inputs, fg_inputs, targets, masks, lr, beta1=0.5, group=None):
    """One iteration of train_pixrefer.py:136-143 on this replica: forward, both backward passes,
    (data parallel: RCCL all-reduce-mean of the two gradient arenas, the discriminator's overlapped
    with the generator backward), Adam(D) then Adam(G)."""
    self.forward(inputs, fg_inputs, targets, masks)
    if group is None and self.fused_update:
      # both passes AND both Adam updates + weight re-packs in one executor call: every arena range is updated as soon as its
      # gradients are final, under the rest of the backward pass (bit-identical to backward() + adam_step())
      self.t_d += 1
      self.t_g += 1
      (m_g, v_g), (m_d, v_d) = self.adam["g"], self.adam["d"]
      _lib.check(self.L.vp_pixrefer_backward_update(self.h, _ptr(m_g), _ptr(v_g), _ptr(m_d), _ptr(v_d), self.t_g, self.t_d, lr, beta1,
                                                    0.999, 1e-8, _stream()), "vp_pixrefer_backward_update")
      return
    if group is None:
      self.backward()          # both passes, the discriminator-loss pass on the executor's side stream
    else:
      # data parallel: every bucket is all-reduced on the communication stream as soon as its stage has run, and its Adam update +
      # weight re-pack follow right behind the collective on that stream - under the stages that still compute
      from .parallel import GradExchange
      if not self._dp_streams_set:
        # a communication stream of the exchange's own is a fifth busy stream: the executor then keeps to three (include/vp_hip.h
        # vp_pixrefer_use_streams).  With the collectives on the executor's side stream (the default) it keeps its fourth: one-rank RCCL,
        # bf16 transport, 16 / 8 / 4 frames: 4.54 / 3.08 / 2.35 ms with three executor streams, 4.45 / 3.00 / 2.29 with four
        # (profiles/r06_exp_dp1_one_rank_rccl.txt)
        if self.dp_own_stream:
          self.use_streams(3)
        self._dp_streams_set = True
      ex = self._exchange
      if ex is None or ex.group is not group or ex.transport != self.grad_transport:
        # the collectives go to the executor's side stream (behind the discriminator-loss pass, where the single-GPU schedule runs its
        # optimiser): one stream fewer competing for the device's hardware queues.  dp_own_stream: a stream of the exchange's own
        sp = None if self.dp_own_stream else self.L.vp_pixrefer_side_stream(self.h)
        ex = self._exchange = GradExchange(group, self.grad_transport, torch.cuda.ExternalStream(sp) if sp else None)
      self.t_d += 1
      self.t_g += 1
      (m_g, v_g), (m_d, v_d) = self.adam["g"], self.adam["d"]

      def update(which, bucket):
        m, v, t = (m_d, v_d, self.t_d) if which else (m_g, v_g, self.t_g)
        return lambda sp: _lib.check(self.L.vp_pixrefer_update_bucket(self.h, which, bucket, _ptr(m), _ptr(v), t, lr, beta1, 0.999, 1e-8, sp),
                                     "vp_pixrefer_update_bucket")
      ex.begin_step()
      _lib.check(self.L.vp_pixrefer_backward_d_fork(self.h, _stream()), "vp_pixrefer_backward_d_fork")
      d_early = not self.dp_own_stream
      for stage, (lo, hi) in enumerate(self.grad_buckets_g()):
        self.backward_g_stage(stage)
        if stage == 0 and d_early:
          # the discriminator's (small) bucket right behind its loss pass: stage 0 has started that pass on the side stream - where the
          # collectives are issued too, so stream order covers it - and holds the last read of the discriminator's weights (the
          # generator-loss pass through D), so the bucket's Adam update + re-pack may follow at once.  As the LAST bucket (rounds 3-5) its
          # all-reduce -> Adam -> re-pack chain was the tail of every data-parallel step
          ex.start(self.grads_d, then=update(1, 0), name="discriminator")
        ex.start(self.grads_g[lo:hi], then=update(0, stage), name="generator stage %d" % stage)
      _lib.check(self.L.vp_pixrefer_backward_d_join(self.h, _stream()), "vp_pixrefer_backward_d_join")
      if not d_early:
        # a communication stream of the exchange's own: the discriminator-loss pass has to be joined first; its bucket goes last
        ex.start(self.grads_d, then=update(1, 0), name="discriminator")       # (an event on this stream covers the joined pass: include/vp_hip.h)
      ex.finish()
      return
    self.adam_step(lr, beta1)

  def adam_step(self, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer on discriminator* then generator* (pixrefer.py:396-407)."""
    self.t_d += 1
    self.t_g += 1
    for key, p, g, t in (("d", self.params_d, self.grads_d, self.t_d), ("g", self.params_g, self.grads_g, self.t_g)):
      m, v = self.adam[key]
      _lib.check(self.L.vp_adam_tf(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), t, lr, beta1, beta2, eps, _stream()),
                 "vp_adam_tf")
    _lib.check(self.L.vp_pixrefer_optimizer_stepped(self.h))      # generator / discriminator moved; the frozen vgg_16 trunk did not

  # ---- named device buffers -----------------------------------------------------------------------
  def tensor(self, name):
    """View of a named buffer inside the workspace (no copy)."""
    p = ctypes.c_void_p()
    shp = (ctypes.c_int64 * 4)()
    dt = ctypes.c_int()
    _lib.check(self.L.vp_pixrefer_tensor(self.h, name.encode(), ctypes.byref(p), shp, ctypes.byref(dt)),
               "vp_pixrefer_tensor(%s)" % name)
    shape = tuple(int(s) for s in shp)
    tdt = torch.bfloat16 if dt.value == VP_BF16 else torch.float32
    nbytes = int(np.prod(shape)) * (2 if dt.value == VP_BF16 else 4)
    off = p.value - self.workspace.data_ptr()
    assert 0 <= off and off + nbytes <= self.workspace.numel(), name
    return self.workspace[off:off + nbytes].view(tdt).view(shape)

  FETCH = {"Outputs": 0, "Outputs_u8": 1, "Alphas": 2, "Outputs_FG": 3}

  def fetch(self, name):
    """A node value of the reference's graphs formed on the device from the last forward pass (vp_pixrefer_fetch): 'Outputs' (deprocessed
    float32), 'Outputs_u8' (the uint8 frames infer_bfmvid.py:243 writes), 'Alphas' (three channels), 'Outputs_FG' (with the
    pixrefer.py:436 quirk on an inference plan).  [N, H, H, 3] device tensor."""
    n, hgt = self.desc.batch, self.desc.height
    out = torch.empty((n, hgt, hgt, 3), dtype=torch.uint8 if name == "Outputs_u8" else torch.float32, device=self.device)
    _lib.check(self.L.vp_pixrefer_fetch(self.h, self.FETCH[name], _ptr(out), _stream()), "vp_pixrefer_fetch(%s)" % name)
    return out

  def profile(self, on):
    """Per-launch HIP-event timing of the conv kernels; while it is on the executor keeps every kernel on one stream (timing a
    kernel that shares the GPU with another stream's kernels measures the sharing, not the kernel)."""
    self.L.vp_profile_enable(int(on))
    self.set_option("overlap", 0 if on else 1)

  def set_option(self, key, value):
    """Schedule option of THIS engine's plan (vp_pixrefer_set_option: "overlap", "d_backward_fork", "d_beside_vgg", "store_first_raw")."""
    _lib.check(self.L.vp_pixrefer_set_option(self.h, key.encode(), int(value)), "vp_pixrefer_set_option(%s)" % key)

  def phase_ms(self):
    """After vp_tune("phase_marks", 1) and a step on a training plan: milliseconds of the step's phases on the caller's stream
    (see include/vp_hip.h, vp_pixrefer_phase_ms)."""
    buf = (ctypes.c_float * 12)()
    n = self.L.vp_pixrefer_phase_ms(self.h, buf, 12)
    return [float(buf[i]) for i in range(n)]

  def profile_collect(self):
    import json
    n = self.L.vp_profile_collect(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    self.L.vp_profile_collect(buf, len(buf))
    return json.loads(buf.value.decode())

  def losses(self):
    l = self.tensor("losses").view(-1).float().cpu().numpy()
    return {"Discrim_loss": float(l[0]), "Gen_loss_GAN": float(l[1]), "Gen_loss_L1": float(l[2]),
            "Gen_loss": float(l[3]), "Perceptual_loss": float(l[4])}
