"""Host-side drivers of the audio front-end executors (libvp_hip.so: vp_logmel_*, vp_bfmnet_*)."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import BfmNetDesc, LogMelDesc


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class LogMel:
  """DataGenerator.extract_mfcc (generator/generator.py:60-80) for a fixed [batch, samples] shape."""

  def __init__(self, batch, samples, sample_rate=16000, num_mel_bins=80, win_length=512, hop_step=128, fft_length=512,
               lower_hz=80.0, upper_hz=7600.0):
    if not torch.cuda.is_available():
      raise RuntimeError("LogMel needs an MI355X (no CPU fallback)")
    self.L = _lib.lib()
    self.desc = LogMelDesc(sample_rate, num_mel_bins, win_length, hop_step, fft_length, lower_hz, upper_hz, batch, samples)
    d = ctypes.byref(self.desc)
    ws = self.L.vp_logmel_workspace_bytes(d)
    if ws == 0:
      raise ValueError("invalid log-mel descriptor (win_length must equal fft_length, samples >= win_length)")
    self.frames = self.L.vp_logmel_frames(d)
    self.workspace = torch.zeros(ws, dtype=torch.uint8, device="cuda")
    h = ctypes.c_void_p()
    _lib.check(self.L.vp_logmel_create(d, _ptr(self.workspace), ws, _stream(), ctypes.byref(h)), "vp_logmel_create")
    self.h = h

  def __call__(self, pcm):
    pcm = pcm.contiguous()
    assert pcm.is_cuda and pcm.dtype == torch.float32 and tuple(pcm.shape) == (self.desc.batch, self.desc.samples)
    out = torch.empty(self.desc.batch, self.frames, self.desc.num_mel_bins, dtype=torch.float32, device=pcm.device)
    _lib.check(self.L.vp_logmel_forward(self.h, _ptr(pcm), _ptr(out), _stream()), "vp_logmel_forward")
    return out

  def __del__(self):
    try:
      if getattr(self, "h", None):
        self.L.vp_logmel_destroy(self.h)
        self.h = None
    except Exception:
      pass


def bfmnet_manifest():
  L = _lib.lib()
  out = []
  name = ctypes.create_string_buffer(256)
  off = ctypes.c_size_t()
  nd = ctypes.c_int()
  shp = (ctypes.c_int64 * 4)()
  i = 0
  while L.vp_bfmnet_param_info(i, name, 256, ctypes.byref(off), ctypes.byref(nd), shp) == 0:
    out.append((name.value.decode(), int(off.value), tuple(int(shp[k]) for k in range(nd.value))))
    i += 1
  return out


class BFMNetEngine:
  """BFMNet.build_inference_op (bfmnet.py:325-333) for a fixed [batch, frames] shape."""

  def __init__(self, batch, frames, num_mel_bins=80, dtype="f32"):
    """dtype "f32": the parity path; "bf16": MfccNet activations / 1x1-conv operands in bf16 (f32 accumulation, f32 head)."""
    if not torch.cuda.is_available():
      raise RuntimeError("BFMNetEngine needs an MI355X (no CPU fallback)")
    self.L = _lib.lib()
    self.desc = BfmNetDesc(batch, frames, num_mel_bins, {"f32": _lib.VP_F32, "bf16": _lib.VP_BF16}[dtype])
    d = ctypes.byref(self.desc)
    self.manifest = bfmnet_manifest()
    self.params = torch.zeros(self.L.vp_bfmnet_param_count(), dtype=torch.float32, device="cuda")
    ws = self.L.vp_bfmnet_workspace_bytes(d)
    if ws == 0:
      raise ValueError("invalid BFMNet descriptor")
    self.workspace = torch.zeros(ws, dtype=torch.uint8, device="cuda")
    h = ctypes.c_void_p()
    _lib.check(self.L.vp_bfmnet_create(d, _ptr(self.workspace), ws, _ptr(self.params), _stream(), ctypes.byref(h)), "vp_bfmnet_create")
    self.h = h

  def load_params(self, params):
    host = self.params.cpu().numpy()
    for name, off, shape in self.manifest:
      if name in params:
        v = np.asarray(params[name], dtype=np.float32)
        assert v.shape == shape, (name, v.shape, shape)
        host[off:off + v.size] = v.reshape(-1)
    self.params.copy_(torch.from_numpy(host))
    _lib.check(self.L.vp_bfmnet_params_changed(self.h))

  def get_params(self):
    host = self.params.cpu().numpy()
    return {name: host[off:off + int(np.prod(shape))].reshape(shape).copy() for name, off, shape in self.manifest}

  def set_decoder_dropout(self, mask0=None, mask1=None):
    """Opt-in (include/vp_hip.h, vp_bfmnet_set_decoder_dropout): masks [B,T,128] / [B,T,64] with entries 0 or 1 / keep_prob multiply the
    decoder's two hidden activations in every following forward - the reference's unconditional tf.nn.dropout (bfmnet.py:114,116);
    None clears them (the deterministic default)."""
    B, T = self.desc.batch, self.desc.frames
    keep = []
    for m, c in ((mask0, 128), (mask1, 64)):
      if m is not None:
        m = m.to(self.params.device, torch.float32).contiguous()
        assert m.numel() == B * T * c, (tuple(m.shape), (B, T, c))
      keep.append(m)
    self._drop_masks = keep              # the executor keeps the raw pointers: the tensors must outlive the forwards
    _lib.check(self.L.vp_bfmnet_set_decoder_dropout(self.h, _ptr(keep[0]), _ptr(keep[1])), "vp_bfmnet_set_decoder_dropout")

  def draw_decoder_dropout(self, rate=0.25, generator=None):
    """One draw of the two masks as tf.nn.dropout(keep_prob = 1 - rate) makes them (from torch's device generator, not TensorFlow's
    stream) and set_decoder_dropout with them."""
    B, T = self.desc.batch, self.desc.frames
    keep = 1.0 - rate
    mk = lambda c: (torch.rand(B, T, c, device=self.params.device, generator=generator) < keep).to(torch.float32) / keep
    m0, m1 = mk(128), mk(64)
    self.set_decoder_dropout(m0, m1)
    return m0, m1

  def forward(self, ears, mfccs, seq_len):
    B, T = self.desc.batch, self.desc.frames
    ears, mfccs = ears.contiguous(), mfccs.contiguous()
    assert tuple(ears.shape) == (B, T, 1) and tuple(mfccs.shape) == (B, 5 * T, self.desc.num_mel_bins)
    seq = torch.as_tensor(seq_len, dtype=torch.int32, device=mfccs.device).contiguous()
    out = torch.empty(B, T, 64, dtype=torch.float32, device=mfccs.device)
    _lib.check(self.L.vp_bfmnet_forward(self.h, _ptr(ears), _ptr(mfccs), _ptr(seq), _ptr(out), _stream()), "vp_bfmnet_forward")
    return out

  def tensor(self, name):
    p = ctypes.c_void_p()
    shp = (ctypes.c_int64 * 4)()
    _lib.check(self.L.vp_bfmnet_tensor(self.h, name.encode(), ctypes.byref(p), shp), "vp_bfmnet_tensor")
    n = int(shp[0] * shp[1] * shp[2])
    off = p.value - self.workspace.data_ptr()
    return self.workspace[off:off + 4 * n].view(torch.float32).view(int(shp[0]), int(shp[1]), int(shp[2]))

  def __del__(self):
    try:
      if getattr(self, "h", None):
        self.L.vp_bfmnet_destroy(self.h)
        self.h = None
    except Exception:
      pass
