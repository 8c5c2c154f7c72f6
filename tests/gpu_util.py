"""Helpers for the -m gpu parity tests: call libvp_hip.so through its C ABI on torch device buffers."""
import ctypes

import numpy as np
import torch

from voicepuppet_amd import _lib
from voicepuppet_amd._lib import ConvDesc, VP_BF16, VP_F32


def ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def tdtype(dtype):
  return torch.bfloat16 if dtype == "bf16" else torch.float32


def to_dev(a, dtype="f32"):
  return torch.tensor(np.asarray(a), dtype=torch.float32).to("cuda").to(tdtype(dtype)).contiguous()


def dev_f32(a):
  return None if a is None else torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda").contiguous()


def rounded(a, dtype):
  """What the device actually sees: float32, or float32 rounded to bf16."""
  t = torch.tensor(np.asarray(a), dtype=torch.float32)
  if dtype == "bf16":
    t = t.to(torch.bfloat16).float()
  return t.numpy().astype(np.float64)


def rel_l2(a, b):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def conv_desc(kind, n, h, w, cin, cout, k, s, p, dtype, in_act=0, out_act=0):
  return ConvDesc(kind, n, h, w, cin, cout, k, s, p, VP_BF16 if dtype == "bf16" else VP_F32, in_act, out_act)


def out_hw(d):
  if d.kind == 1:
    return 2 * d.h, 2 * d.w
  return (d.h + 2 * d.pad - d.ksize) // d.stride + 1, (d.w + 2 * d.pad - d.ksize) // d.stride + 1


def workspace(d):
  n = _lib.lib().vp_conv_workspace_bytes(ctypes.byref(d))
  assert n > 0
  return torch.zeros(n, dtype=torch.uint8, device="cuda")


def conv_fwd(d, x, scale, shift, w, bias, dtype):
  L = _lib.lib()
  ho, wo = out_hw(d)
  y = torch.full((d.n, ho, wo, d.cout), float("nan"), dtype=tdtype(dtype), device="cuda")
  ws = workspace(d)
  xd, wd = to_dev(x, dtype), dev_f32(w)
  sc, sh, bs = dev_f32(scale), dev_f32(shift), dev_f32(bias)
  _lib.check(L.vp_conv_fwd(ctypes.byref(d), ptr(xd), ptr(sc), ptr(sh), ptr(wd), ptr(bs), ptr(y), ptr(ws), stream()), "vp_conv_fwd")
  torch.cuda.synchronize()
  return y.float().cpu().numpy().astype(np.float64)


def conv_bwd_data(d, dy, w, dtype):
  L = _lib.lib()
  dx = torch.full((d.n, d.h, d.w, d.cin), float("nan"), dtype=tdtype(dtype), device="cuda")
  ws = workspace(d)
  dyd, wd = to_dev(dy, dtype), dev_f32(w)
  _lib.check(L.vp_conv_bwd_data(ctypes.byref(d), ptr(dyd), ptr(wd), ptr(dx), ptr(ws), stream()), "vp_conv_bwd_data")
  torch.cuda.synchronize()
  return dx.float().cpu().numpy().astype(np.float64)


def conv_bwd_weight(d, x, scale, shift, dy, wshape, dtype):
  L = _lib.lib()
  dw = torch.full(wshape, float("nan"), dtype=torch.float32, device="cuda")
  ws = workspace(d)
  xd, dyd = to_dev(x, dtype), to_dev(dy, dtype)
  sc, sh = dev_f32(scale), dev_f32(shift)
  _lib.check(L.vp_conv_bwd_weight(ctypes.byref(d), ptr(xd), ptr(sc), ptr(sh), ptr(dyd), ptr(dw), ptr(ws), stream()), "vp_conv_bwd_weight")
  torch.cuda.synchronize()
  return dw.cpu().numpy().astype(np.float64)
