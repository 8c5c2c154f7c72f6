"""Pins the numpy oracle (oracle/pixrefer_ref.py) with independent checks:
direct-loop definitions, finite differences and a torch-CPU float64 autograd
restatement of the same graph (second opinion only - never a product path)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nn_ops as ops
from oracle import pixrefer_ref as ref


def rnd(*shape, seed=0):
  return np.random.default_rng(seed).normal(size=shape)


@pytest.mark.parametrize("stride,pad,k", [(2, 1, 4), (1, 1, 4), (1, 1, 3)])
def test_conv_matches_direct_loops(stride, pad, k):
  x, w, b = rnd(2, 8, 8, 3), rnd(k, k, 3, 5, seed=1), rnd(5, seed=2)
  np.testing.assert_allclose(ops.conv2d_fwd(x, w, b, stride, pad), ops.conv2d_direct(x, w, b, stride, pad), rtol=1e-12, atol=1e-12)


def test_deconv_matches_direct_loops_and_is_conv_input_gradient():
  x, w, b = rnd(2, 5, 5, 3), rnd(4, 4, 6, 3, seed=1), rnd(6, seed=2)
  y = ops.deconv4s2_fwd(x, w, b)
  np.testing.assert_allclose(y, ops.deconv4s2_direct(x, w, b), rtol=1e-12, atol=1e-12)
  # deconv(x, W) == d/d(input) of conv(k4,s2,p1) with HWIO kernel W (same array, no flip)
  dx, _, _ = ops.conv2d_bwd(np.zeros((2, 10, 10, 6)), w, x, 2, 1)
  np.testing.assert_allclose(y - b, dx, rtol=1e-12, atol=1e-12)


def test_conv_deconv_bn_pool_grads_vs_torch():
  x, w, dy = rnd(2, 8, 8, 3), rnd(4, 4, 3, 5, seed=1), rnd(2, 4, 4, 5, seed=3)
  dx, dw, db = ops.conv2d_bwd(x, w, dy, 2, 1)
  xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_()
  wt = torch.tensor(w).permute(3, 2, 0, 1).contiguous().requires_grad_()
  yt = F.conv2d(xt, wt, stride=2, padding=1)
  yt.backward(torch.tensor(dy).permute(0, 3, 1, 2))
  np.testing.assert_allclose(dx, xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-12)
  np.testing.assert_allclose(dw, wt.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-10, atol=1e-12)

  x, w, dy = rnd(2, 4, 4, 3), rnd(4, 4, 5, 3, seed=1), rnd(2, 8, 8, 5, seed=3)
  dx, dw, db = ops.deconv4s2_bwd(x, w, dy)
  xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_()
  wt = torch.tensor(w).permute(3, 2, 0, 1).contiguous().requires_grad_()   # HWOI -> [Cin,Cout,kh,kw]
  yt = F.conv_transpose2d(xt, wt, stride=2, padding=1)
  np.testing.assert_allclose(ops.deconv4s2_fwd(x, w, None), yt.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-12)
  yt.backward(torch.tensor(dy).permute(0, 3, 1, 2))
  np.testing.assert_allclose(dx, xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-12)
  np.testing.assert_allclose(dw, wt.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-10, atol=1e-12)

  y, g, bta, dz = rnd(3, 4, 4, 6), rnd(6, seed=5), rnd(6, seed=6), rnd(3, 4, 4, 6, seed=7)
  z, cache = ops.bn_train_fwd(y, g, bta)
  dyo, dg, dbt = ops.bn_train_bwd(dz, cache)
  yt = torch.tensor(y).permute(0, 3, 1, 2).requires_grad_()
  gt, bt = torch.tensor(g).requires_grad_(), torch.tensor(bta).requires_grad_()
  zt = F.batch_norm(yt, None, None, gt, bt, training=True, eps=1e-5)
  np.testing.assert_allclose(z, zt.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-12)
  zt.backward(torch.tensor(dz).permute(0, 3, 1, 2))
  np.testing.assert_allclose(dyo, yt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-9, atol=1e-12)
  np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=1e-10)
  np.testing.assert_allclose(dbt, bt.grad.numpy(), rtol=1e-10)

  x, dy = rnd(2, 6, 6, 3), rnd(2, 3, 3, 3, seed=2)
  y, idx = ops.maxpool2x2_fwd(x)
  xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_()
  yt = F.max_pool2d(xt, 2)
  np.testing.assert_allclose(y, yt.detach().permute(0, 2, 3, 1).numpy())
  yt.backward(torch.tensor(dy).permute(0, 3, 1, 2))
  np.testing.assert_allclose(ops.maxpool2x2_bwd(dy, idx, x.shape), xt.grad.permute(0, 2, 3, 1).numpy())


def test_bn_zero_variance_edge_case():
  # N=1 and a 1x1 bottleneck (H=256, m5): x - mu == 0  =>  y == beta exactly
  y = rnd(1, 1, 1, 8)
  z, _ = ops.bn_train_fwd(y, rnd(8, seed=1), np.arange(8.0))
  np.testing.assert_array_equal(z.reshape(-1), np.arange(8.0))


def test_manifest_parameter_counts():
  g, d = ref.param_manifest(64, 64)
  assert sum(int(np.prod(s)) for _, s in g) == 35158852
  assert sum(int(np.prod(s)) for _, s in d) == 2769601


# ---------------------------------------------------------------------------
# torch restatement of the whole training graph (float64 autograd): oracle/pixrefer_torch.py
# ---------------------------------------------------------------------------
from oracle.pixrefer_torch import TorchGraph, torch_graph  # noqa: E402


def synth_batch(n, h, seed=0):
  rng = np.random.default_rng(seed)
  return (rng.uniform(size=(n, h, h, 6)), rng.uniform(size=(n, h, h, 6)),
          rng.uniform(size=(n, h, h, 3)), rng.uniform(size=(n, h, h, 3)))


@pytest.fixture(scope="module")
def mini():
  ngf = ndf = 4
  p = ref.init_params(ngf, ndf, seed=1)
  # non-zero biases / betas so that every term is exercised
  rng = np.random.default_rng(5)
  for k in p:
    if k.endswith('bias') or k.endswith('beta'):
      p[k] = rng.normal(0, 0.1, p[k].shape)
  batch = synth_batch(2, 256, seed=2)
  nodes = ref.forward_backward(p, *batch, ngf=ngf, ndf=ndf)
  return p, batch, nodes, ngf, ndf


def test_full_graph_losses_and_grads_vs_torch(mini):
  p, batch, nodes, ngf, ndf = mini
  tg = torch_graph(p, *batch, ngf, ndf)
  assert nodes['Discrim_loss'] == pytest.approx(tg['d_loss'], rel=1e-10)
  assert nodes['Gen_loss_GAN'] == pytest.approx(tg['g_gan'], rel=1e-10)
  assert nodes['Gen_loss_L1'] == pytest.approx(tg['g_l1'], rel=1e-10)
  assert nodes['Perceptual_loss'] == pytest.approx(tg['content'], rel=1e-10)
  np.testing.assert_allclose(nodes['Outputs_raw'], tg['outputs'], rtol=1e-9, atol=1e-12)
  for grads, tgr in ((nodes['Discrim_grads'], tg['dgr']), (nodes['Gen_grads'], tg['ggr'])):
    for k, g in tgr.items():
      o = grads[k]
      scale = max(np.abs(g).max(), 1e-30)
      if k.endswith('bias') and np.all(o == 0):   # analytically-zero bias grads are pinned to 0
        assert np.abs(g).max() < 1e-9 * max(1.0, np.abs(grads[k.replace('bias', 'kernel')]).max()), k
        continue
      assert np.abs(o - g).max() / scale < 1e-7, k


def test_adam_and_lr_schedule(mini):
  p, batch, nodes, ngf, ndf = mini
  st = ref.TrainState({k: v.copy() for k, v in p.items()}, ngf, ndf)
  before = {k: v.copy() for k, v in st.p.items()}
  out = st.step(*batch)
  assert out['Global_step'] == 2 and out['Lr'] == pytest.approx(3e-4)
  # first TF-Adam step: |delta| = lr_t * |g|/(|g|+eps') ~ lr  (sign of the gradient)
  k = 'discriminator/layer_4/conv2d/kernel'
  g = out['Discrim_grads'][k]
  lr_t = 3e-4 * np.sqrt(1 - 0.999) / (1 - 0.5)
  expect = before[k] - lr_t * (0.5 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
  np.testing.assert_allclose(st.p[k], expect, rtol=1e-12, atol=0)
  assert ref.learning_rate(3e-4, 1999, 1000, 0.999) == pytest.approx(3e-4 * 0.999)
  assert ref.learning_rate(3e-4, 2000, 1000, 0.999) == pytest.approx(3e-4 * 0.999 ** 2)


def test_finite_difference_generator_weight(mini):
  p, batch, nodes, ngf, ndf = mini
  k = 'generator/merged_decoder_3/conv2d_transpose/kernel'
  idx = (1, 2, 3, 5)
  h = 1e-5
  vals = []
  for s in (+1, -1):
    q = {a: b.copy() for a, b in p.items()}
    q[k][idx] += s * h
    vals.append(ref.forward_backward(q, *batch, ngf=ngf, ndf=ndf, want_grads=False)['Gen_loss'])
  fd = (vals[0] - vals[1]) / (2 * h)
  assert nodes['Gen_grads'][k][idx] == pytest.approx(fd, rel=2e-4, abs=1e-9)


def test_three_adam_steps_numpy_oracle_vs_torch_autograd(mini):
  """TrainState.step (hand-written backward + TF-Adam) against autograd + the same update rule, 3 consecutive steps."""
  p, batch, nodes, ngf, ndf = mini
  st = ref.TrainState({k: v.copy() for k, v in p.items()}, ngf, ndf)
  tg = TorchGraph(p, ngf, ndf)
  for _ in range(3):
    out = st.step(*batch)
    got = tg.step(*batch)
    assert out['Discrim_loss'] == pytest.approx(got['d_loss'], rel=1e-9)
    assert out['Gen_loss'] == pytest.approx(got['g_loss'], rel=1e-9)
  assert st.global_step == tg.global_step == 6
  for k in tg.g_names + tg.d_names:
    a, b = st.p[k], tg.tp[k].detach().numpy()
    if k.endswith('bias') and np.all(a == p[k]):
      continue   # analytically-zero bias gradients: the oracle pins them to 0, autograd leaves round-off that Adam's sign-like first steps amplify
    assert np.abs(a - b).max() <= 1e-6 * max(np.abs(a).max(), 1e-3), k
