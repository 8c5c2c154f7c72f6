"""Oracle primitives: NHWC conv / transposed conv / batch-norm / pooling, fwd + bwd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function restates the
TF1.x op the reference calls; TF itself is not vendored in /root/reference, so
the semantics are stated from the TF 1.14 API contract and cross-checked in
tests/ (direct-loop definitions, finite differences, torch-CPU float64).

All tensors are NHWC numpy arrays; the working dtype follows the inputs
(float64 for parity, float32 for the cpu_baseline timing leg).
"""
import numpy as np


# --------------------------------------------------------------------------
# conv2d, kernel HWIO, symmetric zero padding `pad`, stride `s`
#   reference call sites: tf.layers.conv2d(k=4, s=2, "same")  pixrefer.py:73
#                         tf.pad 1 + conv2d(k=4, s, "valid")     pixrefer.py:61-64
#                         slim.conv2d 3x3 s1 SAME                vgg_simple.py:138-151
#   SAME with k=4,s=2 on an even size == pad 1/1; 3x3 s1 SAME == pad 1/1.
# --------------------------------------------------------------------------
def _pad_hw(x, pad):
  if pad == 0:
    return x
  return np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))


def conv2d_fwd(x, w, b, stride, pad):
  n, h, wd, cin = x.shape
  kh, kw, _, cout = w.shape
  ho = (h + 2 * pad - kh) // stride + 1
  wo = (wd + 2 * pad - kw) // stride + 1
  xp = _pad_hw(x, pad)
  y = np.zeros((n, ho, wo, cout), dtype=x.dtype)
  for i in range(kh):
    for j in range(kw):
      xs = xp[:, i:i + stride * ho:stride, j:j + stride * wo:stride, :]
      y += xs @ w[i, j]
  if b is not None:
    y += b
  return y


def conv2d_bwd(x, w, dy, stride, pad, need_dx=True, need_dw=True):
  """Returns (dx, dw, db) of conv2d_fwd."""
  n, h, wd, cin = x.shape
  kh, kw, _, cout = w.shape
  _, ho, wo, _ = dy.shape
  xp = _pad_hw(x, pad)
  dw = np.zeros_like(w) if need_dw else None
  dxp = np.zeros_like(xp) if need_dx else None
  dy2 = dy.reshape(-1, cout)
  for i in range(kh):
    for j in range(kw):
      sl = (slice(None), slice(i, i + stride * ho, stride), slice(j, j + stride * wo, stride), slice(None))
      if need_dw:
        dw[i, j] = xp[sl].reshape(-1, cin).T @ dy2
      if need_dx:
        dxp[sl] += dy @ w[i, j].T
  dx = None
  if need_dx:
    dx = dxp[:, pad:pad + h, pad:pad + wd, :] if pad else dxp
  db = dy2.sum(axis=0)
  return dx, dw, db


# --------------------------------------------------------------------------
# conv2d_transpose k=4 s=2 "same", kernel HWOI = [4,4,Cout,Cin]
#   reference: tf.layers.conv2d_transpose  pixrefer.py:85
#   == gradient of a k4/s2/pad1 conv w.r.t. its input:
#      out[n, 2i+kh-1, 2j+kw-1, co] += in[n,i,j,ci] * W[kh,kw,co,ci]
# --------------------------------------------------------------------------
def deconv4s2_fwd(x, w, b):
  n, h, wd, cin = x.shape
  kh, kw, cout, _ = w.shape
  yp = np.zeros((n, 2 * h + 2, 2 * wd + 2, cout), dtype=x.dtype)
  for i in range(kh):
    for j in range(kw):
      yp[:, i:i + 2 * h:2, j:j + 2 * wd:2, :] += x @ w[i, j].T
  y = yp[:, 1:1 + 2 * h, 1:1 + 2 * wd, :]
  if b is not None:
    y = y + b
  return np.ascontiguousarray(y)


def deconv4s2_bwd(x, w, dy, need_dx=True):
  """Returns (dx, dw, db) of deconv4s2_fwd."""
  n, h, wd, cin = x.shape
  kh, kw, cout, _ = w.shape
  dyp = np.pad(dy, ((0, 0), (1, 1), (1, 1), (0, 0)))
  dw = np.zeros_like(w)
  dx = np.zeros_like(x) if need_dx else None
  x2 = x.reshape(-1, cin)
  for i in range(kh):
    for j in range(kw):
      g = dyp[:, i:i + 2 * h:2, j:j + 2 * wd:2, :]
      dw[i, j] = g.reshape(-1, cout).T @ x2
      if need_dx:
        dx += g @ w[i, j]
  db = dy.reshape(-1, cout).sum(axis=0)
  return dx, dw, db


# --------------------------------------------------------------------------
# batch norm, permanently in training mode (pixrefer.py:99-101):
#   statistics over (N,H,W) of the current batch, biased variance, eps in sqrt
# --------------------------------------------------------------------------
def bn_train_fwd(y, gamma, beta, eps=1e-5):
  mu = y.mean(axis=(0, 1, 2))
  var = ((y - mu) ** 2).mean(axis=(0, 1, 2))
  rstd = 1.0 / np.sqrt(var + eps)
  xhat = (y - mu) * rstd
  return gamma * xhat + beta, (xhat, rstd, gamma)


def bn_train_bwd(dz, cache):
  xhat, rstd, gamma = cache
  dgamma = (dz * xhat).sum(axis=(0, 1, 2))
  dbeta = dz.sum(axis=(0, 1, 2))
  m = dz.shape[0] * dz.shape[1] * dz.shape[2]
  dy = gamma * rstd * (dz - dbeta / m - xhat * (dgamma / m))
  return dy, dgamma, dbeta


# --------------------------------------------------------------------------
# activations (pixrefer.py:88-97, 243, 274, 131)
# --------------------------------------------------------------------------
def lrelu(x, a=0.2):
  return (0.5 * (1 + a)) * x + (0.5 * (1 - a)) * np.abs(x)


def lrelu_grad(x, a=0.2):
  # d/dx [c1*x + c2*|x|] with tf.abs'(0) = sign(0) = 0
  return (0.5 * (1 + a)) + (0.5 * (1 - a)) * np.sign(x)


def relu(x):
  return np.maximum(x, 0)


def relu_grad(x):
  return (x > 0).astype(x.dtype)


def sigmoid(x):
  return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# 2x2/s2 max pool (slim.max_pool2d, vgg_simple.py:143,150).  Gradient goes to
# the arg-max; on ties TF's MaxPoolGrad routes to the FIRST maximum in window
# scan order (row-major), which is what argmax over the flattened window does.
# --------------------------------------------------------------------------
def maxpool2x2_fwd(x):
  n, h, w, c = x.shape
  xw = x.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4, c)
  idx = xw.argmax(axis=3)
  y = np.take_along_axis(xw, idx[:, :, :, None, :], axis=3)[:, :, :, 0, :]
  return y, idx


def maxpool2x2_bwd(dy, idx, in_shape):
  n, h, w, c = in_shape
  dxw = np.zeros((n, h // 2, w // 2, 4, c), dtype=dy.dtype)
  np.put_along_axis(dxw, idx[:, :, :, None, :], dy[:, :, :, None, :], axis=3)
  return dxw.reshape(n, h // 2, w // 2, 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h, w, c)


# --------------------------------------------------------------------------
# direct-loop definitions (slow; used by tests on toy sizes only)
# --------------------------------------------------------------------------
def conv2d_direct(x, w, b, stride, pad):
  n, h, wd, cin = x.shape
  kh, kw, _, cout = w.shape
  ho = (h + 2 * pad - kh) // stride + 1
  wo = (wd + 2 * pad - kw) // stride + 1
  y = np.zeros((n, ho, wo, cout), dtype=np.float64)
  for a in range(n):
    for oh in range(ho):
      for ow in range(wo):
        for i in range(kh):
          for j in range(kw):
            ih, iw = oh * stride + i - pad, ow * stride + j - pad
            if 0 <= ih < h and 0 <= iw < wd:
              y[a, oh, ow] += x[a, ih, iw] @ w[i, j]
  return y + (0 if b is None else b)


def deconv4s2_direct(x, w, b):
  n, h, wd, cin = x.shape
  kh, kw, cout, _ = w.shape
  y = np.zeros((n, 2 * h, 2 * wd, cout), dtype=np.float64)
  for a in range(n):
    for i in range(h):
      for j in range(wd):
        for p in range(kh):
          for q in range(kw):
            oh, ow = 2 * i + p - 1, 2 * j + q - 1
            if 0 <= oh < 2 * h and 0 <= ow < 2 * wd:
              y[a, oh, ow] += w[p, q] @ x[a, i, j]
  return y + (0 if b is None else b)
