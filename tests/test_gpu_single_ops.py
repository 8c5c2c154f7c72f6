"""Parity of the single pointwise / audio entry points (include/vp_hip.h, SURVEY.md 8b list) against the float64 oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import audio_ref as ar
from oracle import nn_ops as ops
from oracle import pixrefer_ref as ref
from voicepuppet_amd import _lib
from voicepuppet_amd._lib import VP_BF16, VP_F32

import gpu_util as gu

pytestmark = pytest.mark.gpu
P = gu.ptr


def dev(a, dt=torch.float32):
  return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda").to(dt).contiguous()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_maxpool2x2_fwd_bwd(dtype):
  L = _lib.lib()
  rng = np.random.default_rng(0)
  x = np.maximum(gu.rounded(rng.normal(size=(2, 8, 12, 16)), dtype), 0)      # the pooled tensor is a conv + relu output (vgg_simple.py:138-146)
  td, code = gu.tdtype(dtype), (VP_BF16 if dtype == "bf16" else VP_F32)
  xd = dev(x, td)
  y = torch.empty(2, 4, 6, 16, dtype=td, device="cuda")
  _lib.check(L.vp_maxpool2x2_fwd(P(xd), P(y), 2, 8, 12, 16, code, gu.stream()))
  yr, idx = ops.maxpool2x2_fwd(x)
  assert np.array_equal(y.float().cpu().numpy(), yr)                 # a max of stored values: exact
  dy = gu.rounded(rng.normal(size=yr.shape), dtype)
  dx = torch.full((2, 8, 12, 16), float("nan"), dtype=td, device="cuda")
  dyd = dev(dy, td)                      # (device temporaries must outlive the asynchronous call: keep them in variables)
  _lib.check(L.vp_maxpool2x2_bwd(P(xd), P(dyd), P(dx), 2, 8, 12, 16, code, gu.stream()))
  # backward goes through the pool AND the relu that produced x: nothing flows into a window whose maximum is 0
  assert np.array_equal(dx.float().cpu().numpy(), ops.maxpool2x2_bwd(dy, idx, x.shape) * (x > 0))


def test_composite_fwd():
  L = _lib.lib()
  rng = np.random.default_rng(1)
  y4 = rng.normal(0, 1.5, size=(2, 16, 16, 4))
  tg = rng.uniform(size=(2, 16, 16, 3))
  o4, out, fg = [torch.empty(2, 16, 16, c, device="cuda") for c in (4, 3, 3)]
  y4d, tgd = dev(y4), dev(tg)
  _lib.check(L.vp_composite_fwd(P(y4d), P(tgd), P(o4), P(out), P(fg), 2, 256, gu.stream()))
  t4 = np.tanh(np.float32(y4).astype(np.float64))
  want_out, _, want_fg = ref.composite(t4, np.float32(tg).astype(np.float64) * 2 - 1)
  assert gu.rel_l2(o4.cpu().numpy(), t4) < 1e-6
  assert gu.rel_l2(out.cpu().numpy(), want_out) < 1e-6 and gu.rel_l2(fg.cpu().numpy(), want_fg) < 1e-6


def test_gan_loss_and_seeds():
  L = _lib.lib()
  m, gw = 450, 1.0
  lg = np.float32(np.random.default_rng(2).normal(0, 2, size=(3, m))).astype(np.float64)
  sd = torch.zeros(3, m, 8, device="cuda"); sg = torch.zeros(m, 8, device="cuda")
  pred = torch.empty(2, m, device="cuda"); losses = torch.zeros(8, device="cuda")
  lgd = dev(lg)
  _lib.check(L.vp_gan_loss(P(lgd), P(sd), P(sg), P(pred), P(losses), m, ctypes.c_float(gw), VP_F32, gu.stream()))
  p = 1 / (1 + np.exp(-lg))
  pr, pf, eps = (p[0] + p[1]) / 2, p[2], 1e-12
  d_loss = np.mean(-(2 * np.log(pr + eps) + np.log(1 - pf + eps)))          # pixrefer.py:334-341 (the factor 2 is the reference's)
  g_loss = np.mean(-np.log(pf + eps))                                        # :346
  got = losses.cpu().numpy()
  assert abs(got[0] - d_loss) < 1e-5 * abs(d_loss) and abs(got[1] - g_loss) < 1e-5 * abs(g_loss)
  assert gu.rel_l2(pred.cpu().numpy(), np.stack([pr, pf])) < 1e-6
  dpr = -2 / (pr + eps) / m * 0.5
  want_d = np.stack([dpr * p[0] * (1 - p[0]), dpr * p[1] * (1 - p[1]), 1 / (1 - pf + eps) / m * pf * (1 - pf)])
  want_g = gw * (-1 / (pf + eps)) / m * pf * (1 - pf)
  assert gu.rel_l2(sd[..., 0].cpu().numpy(), want_d) < 1e-5 and gu.rel_l2(sg[..., 0].cpu().numpy(), want_g) < 1e-5
  assert not sd[..., 1:].any() and not sg[..., 1:].any()


def test_dwconv7x3_bn_act():
  L = _lib.lib()
  rng = np.random.default_rng(3)
  x = np.float32(rng.normal(size=(2, 9, 6, 32))).astype(np.float64)
  w = np.float32(rng.normal(0, 0.3, size=(7, 3, 32, 1))).astype(np.float64)
  b = np.float32(rng.normal(0, 0.5, size=32)).astype(np.float64)
  y = torch.empty(2, 9, 6, 32, device="cuda")
  xd, wd, bd = dev(x), dev(w.reshape(21, 32)), dev(b)
  _lib.check(L.vp_dwconv7x3_bn_act(P(xd), P(wd), P(bd), P(y), 2, 9, 6, 32, gu.stream()))
  assert gu.rel_l2(y.cpu().numpy(), ar.relu6(ar.depthwise_same(x, w) + b)) < 1e-6


@pytest.mark.parametrize("k,s,shape", [((2, 2), (1, 2), (2, 7, 5, 16)), ((5, 3), (5, 3), (1, 25, 3, 8)), ((2, 2), (1, 2), (1, 4, 10, 4))])
def test_maxpool_hw_same(k, s, shape):
  L = _lib.lib()
  x = np.float32(np.random.default_rng(4).normal(size=shape)).astype(np.float64)
  want = ar.maxpool_same(x, k, s)
  y = torch.empty(*want.shape, device="cuda")
  xd = dev(x)
  _lib.check(L.vp_maxpool_hw(P(xd), P(y), shape[0], shape[1], shape[2], shape[3], k[0], k[1], s[0], s[1], gu.stream()))
  assert np.array_equal(y.cpu().numpy().astype(np.float64), want)


def test_gru_seq():
  L = _lib.lib()
  B, T, I, H = 4, 7, 64, 256
  rng = np.random.default_rng(5)
  f = lambda *sh: np.float32(rng.normal(0, 0.08, size=sh)).astype(np.float64)
  x, wg, bg, wc, bc = f(B, T, I) * 10, f(I + H, 2 * H), f(2 * H) + 1.0, f(I + H, H), f(H)
  seq = np.array([7, 4, 1, 0], np.int32)                        # ragged, down to an empty sequence (dynamic_rnn: all-zero outputs)
  want = ar.gru_seq(x, seq, wg, bg, wc, bc)
  xg, xc = x @ wg[:I] + bg, x @ wc[:I] + bc                       # the input halves of the two kernels are plain GEMMs (vp_conv_fwd 1x1)
  out = torch.full((B, T, H), float("nan"), device="cuda")
  xgd, xcd, whg, whc, sq = dev(xg), dev(xc), dev(wg[I:]), dev(wc[I:]), torch.tensor(seq, device="cuda")
  _lib.check(L.vp_gru_seq(P(xgd), P(xcd), P(whg), P(whc), P(sq), P(out), B, T, gu.stream()))
  assert gu.rel_l2(out.cpu().numpy(), want) < 1e-5
  got = out.cpu().numpy()
  assert np.all(got[3] == 0) and np.all(got[2, 1:] == 0) and np.all(want[3] == 0)
