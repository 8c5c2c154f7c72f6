"""Oracle for the bf16 device path: the SAME graph as pixrefer_ref.py, restated in the device's dataflow with a
rounding hook `q` at every point where the HIP path stores a tensor in its compute dtype.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  With q = identity this module reproduces pixrefer_ref.py to
round-off (tests/test_oracle_lowp.py) - that pins the restated dataflow; with q = round-to-bf16 it predicts what a
correct bf16 implementation must produce, INCLUDING which ReLU / leaky-ReLU masks it takes (the masks come from the
rounded activations), so gradients can be compared tightly instead of through ~5 %/layer mask-flip noise.

Rounding points: packed inputs; every stored conv output (after bias / epilogue activation,
except the f32 generator output and discriminator logits); packed weights; materialised act(scale*y+shift);
Outputs_FG where it enters the discriminator / VGG batches; every stored gradient tensor (loss seeds, backward-data
results incl. the read-modify-write accumulation over skip consumers, batch-norm backward output).  Accumulation,
batch-norm statistics, losses and weight gradients are not rounded (f32/f64 on the device).

`hi` (round 4): generator tensors the device keeps in float32 (batch-normalised few-pixel tensors, <= 256 pixels: DESIGN.md "few-pixel
tensors stay float32"): their raw output and their accumulated gradient are rounded to float32, not to the compute dtype; the
materialised activations and the batch-norm backward's result (both MFMA operands) are still rounded to the compute dtype.
"""
import numpy as np

from . import nn_ops as ops
from . import pixrefer_ref as ref


def round_bf16(x):
  """float -> nearest bfloat16 (ties to even), returned as float64."""
  f = np.ascontiguousarray(x, dtype=np.float32)
  u = f.view(np.uint32)
  r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
  return r.view(np.float32).astype(np.float64)


def f32(x):
  return np.asarray(x, dtype=np.float32).astype(np.float64)


IDENT = lambda x: np.asarray(x, dtype=np.float64)
ACT = {'lrelu': (lambda v: ops.lrelu(v, 0.2)), 'relu': ops.relu}
ACT_GRAD = {'lrelu': (lambda v: ops.lrelu_grad(v, 0.2)), 'relu': ops.relu_grad}


def _bn_affine(y, gamma, beta, groups, eps=1e-5):
  """Per-group training-mode statistics of the STORED tensor -> (scale, shift, mean, rstd), each [groups, C]."""
  n = y.shape[0] // groups
  sc, sh, mu, rs = [], [], [], []
  for g in range(groups):
    yy = y[g * n:(g + 1) * n]
    m = yy.mean(axis=(0, 1, 2))
    v = np.maximum((yy * yy).mean(axis=(0, 1, 2)) - m * m, 0)
    r = f32(1.0 / np.sqrt(v + eps))
    a = np.where(v == 0, 0.0, f32(gamma * r))
    b = np.where(v == 0, beta, f32(beta - m * a))
    sc.append(a); sh.append(b); mu.append(f32(m)); rs.append(r)
  return np.array(sc), np.array(sh), np.array(mu), np.array(rs)


def _per_group(y, arr, groups):
  """Broadcast [groups, C] statistics over the batch axis of y."""
  n = y.shape[0] // groups
  return np.repeat(arr, n, axis=0)[:, None, None, :]


class Net(object):
  """One sub-network (generator / discriminator) in device dataflow."""

  def __init__(self, spec, params, prefix, q, groups=1, hi=()):
    self.spec, self.p, self.prefix, self.q, self.groups, self.hi = spec, params, prefix, q, groups, frozenset(hi)
    self.y, self.xa, self.bn, self.dy_log = {}, {}, {}, {}

  def wname(self, scope, kind):
    return '%s/%s/%s/kernel' % (self.prefix, scope, 'conv2d' if kind == 'conv' else 'conv2d_transpose')

  def forward(self, inputs, y_override=None):
    """y_override {scope: stored tensor}: after computing a layer from the (possibly overridden) tensors before it, record the
    relative L2 distance to the override in self.fwd_err and continue from the override (layer-by-layer 'teacher forcing')."""
    q = self.q
    self.fwd_err = {}
    self.y.update(inputs)
    need = {}
    for scope, kind, srcs, cout, bn, pre, stride, final in self.spec:
      for s in srcs:
        if pre is not None:
          need.setdefault(s, set()).add(pre)
    for scope, kind, srcs, cout, bn, pre, stride, final in self.spec:
      xs = [self.y[s] if pre is None else self.xa[(s, pre)] for s in srcs]
      x = xs[0] if len(xs) == 1 else np.concatenate(xs, axis=3)
      w = q(self.p[self.wname(scope, kind)])
      bias = None if bn else f32(self.p[self.wname(scope, kind).replace('kernel', 'bias')])
      y = ops.conv2d_fwd(x, w, bias, stride, 1) if kind == 'conv' else ops.deconv4s2_fwd(x, w, bias)
      y = f32(y) if (final or scope in self.hi) else q(y)            # thin f32 outputs / float32 few-pixel tensors are not rounded
      if y_override is not None and scope in y_override:
        o = np.asarray(y_override[scope], dtype=np.float64)
        self.fwd_err[scope] = float(np.linalg.norm(y - o) / max(np.linalg.norm(o), 1e-30))
        y = o
      self.y[scope] = y
      if bn:
        g = f32(self.p['%s/%s/batch_normalization/gamma' % (self.prefix, scope)])
        b = f32(self.p['%s/%s/batch_normalization/beta' % (self.prefix, scope)])
        self.bn[scope] = _bn_affine(y, g, b, self.groups) + (g,)
      for a in need.get(scope, ()):
        z = y
        if bn:
          sc, sh = self.bn[scope][0], self.bn[scope][1]
          z = _per_group(y, sc, self.groups) * y + _per_group(y, sh, self.groups)
        self.xa[(scope, a)] = q(ACT[a](z))
    return self.y[self.spec[-1][0]]

  def backward(self, dy_last, sub=None, want_dw=True):
    """dy_last: stored gradient w.r.t. the raw output of the last layer.  sub = (sample slice, group index) restricts
    the pass to one batch-norm group (the discriminator's fake application for the generator loss)."""
    q = self.q
    sl, gi = (slice(None), None) if sub is None else sub
    groups = self.groups if sub is None else 1
    dz = {self.spec[-1][0]: dy_last}
    grads = {}
    din = {}
    for scope, kind, srcs, cout, bn, pre, stride, final in reversed(self.spec):
      d = dz[scope]
      if bn:
        sc, sh, mu, rs, gamma = self.bn[scope]
        if gi is not None:
          mu, rs = mu[gi:gi + 1], rs[gi:gi + 1]
        y = self.y[scope][sl]
        zh = (y - _per_group(y, mu, groups)) * _per_group(y, rs, groups)
        n = y.shape[0] // groups
        c1 = np.array([d[g * n:(g + 1) * n].mean(axis=(0, 1, 2)) for g in range(groups)])
        c2 = np.array([(d[g * n:(g + 1) * n] * zh[g * n:(g + 1) * n]).mean(axis=(0, 1, 2)) for g in range(groups)])
        if want_dw:
          grads['%s/%s/batch_normalization/gamma' % (self.prefix, scope)] = (d * zh).sum(axis=(0, 1, 2))
          grads['%s/%s/batch_normalization/beta' % (self.prefix, scope)] = d.sum(axis=(0, 1, 2))
        d = q(gamma * _per_group(y, rs, groups) * (d - _per_group(y, f32(c1), groups) - zh * _per_group(y, f32(c2), groups)))
      xs = [(self.y[s] if pre is None else self.xa[(s, pre)])[sl] for s in srcs]
      x = xs[0] if len(xs) == 1 else np.concatenate(xs, axis=3)
      wn = self.wname(scope, kind)
      w = q(self.p[wn])
      if kind == 'conv':
        dx, dw, db = ops.conv2d_bwd(x, w, d, stride, 1, need_dx=True, need_dw=want_dw)
      else:
        dx, dw, db = ops.deconv4s2_bwd(x, w, d)
      if want_dw:
        grads[wn] = dw
        grads[wn.replace('kernel', 'bias')] = np.zeros_like(db) if bn else db
      c0 = 0
      for s, xv in zip(srcs, xs):
        c = xv.shape[3]
        g = dx[..., c0:c0 + c]
        c0 += c
        if pre is None:          # network input: raw gradient
          din[s] = q(g)
          continue
        g = g * ACT_GRAD[pre](xv)                  # masks from the sign of the STORED activation
        qa = f32 if s in self.hi else q            # a float32 few-pixel tensor accumulates its gradient in float32
        dz[s] = qa(g) if s not in dz else qa(dz[s] + g)   # read-modify-write accumulation over skip consumers
      self.dy_log[scope] = d
    return grads, din


def _gspec(ngf):
  out = []
  for scope, kind, srcs, cout, bn, pre in ref.generator_spec(ngf):
    out.append((scope, kind, srcs, cout, bn, pre, 2, scope == 'decoder_1'))
  return out


def _dspec(ndf):
  out, prev = [], 'd_inputs'
  for scope, cout, stride, bn in ref.discriminator_spec(ndf):
    out.append((scope, 'conv', [prev], cout, bn, None if scope == 'layer_1' else 'lrelu', stride, scope == 'layer_5'))
    prev = scope
  return out


def forward_backward(p, inputs, fg_inputs, targets, masks, ngf=64, ndf=64, l1_weight=500.0, gan_weight=1.0, q=round_bf16,
                     out4_override=None, g_override=None, d_override=None, hi=()):
  """Same contract as pixrefer_ref.forward_backward (inputs in [0,1]); q = IDENT gives the float64 graph.
  out4_override: use this generator output (post-tanh, [N,H,H,4]) for everything downstream of the generator.  The
  generator's own bottleneck (batch-norm over N*1*1 .. N*4*4 values) amplifies single-ulp differences chaotically, so a
  tight check of the discriminator / VGG / loss / composite backward feeds both sides the SAME generator output.
  d_override {scope: stored tensor of the 3N discriminator batch}: the same layer-by-layer teacher forcing for the discriminator
  (errors in D.fwd_err), so that both of its backward passes run on identical activations - one-ulp differences of a stored
  activation otherwise move the batch-norm backward of this 8-channel test net by several per cent."""
  N = inputs.shape[0]
  inp, fg, tgt = f32(f32(inputs) * 2 - 1), f32(f32(fg_inputs) * 2 - 1), f32(f32(targets) * 2 - 1)
  masks = f32(masks)
  G = Net(_gspec(ngf), p, 'generator', q, hi=hi)
  y4 = G.forward({'inputs': q(inp), 'fg_inputs': q(fg[..., :3])}, g_override)
  out4 = np.tanh(y4) if out4_override is None else f32(out4_override)
  outputs, alphas, outputs_fg = ref.composite(out4, tgt)
  ofg_q = q(outputs_fg)
  D = Net(_dspec(ndf), p, 'discriminator', q, groups=3)
  d_in = np.concatenate([np.concatenate([q(inp[..., 3:]), q(fg[..., 3:])], 3), np.concatenate([q(inp[..., :3]), q(fg[..., :3])], 3),
                         np.concatenate([q(inp[..., 3:]), ofg_q], 3)], axis=0)
  logits = D.forward({'d_inputs': d_in}, d_override)
  pr = ops.sigmoid(logits)
  p0, p1, pf = pr[:N], pr[N:2 * N], pr[2 * N:]
  predict_real = (p0 + p1) / 2
  eps = 1e-12
  M = predict_real.size
  discrim_loss = np.mean(-(np.log(predict_real + eps) * 2 + np.log(1 - pf + eps)))
  gen_loss_gan = np.mean(-np.log(pf + eps))
  # VGG trunk on [real | fake]; each record: (name, conv input, stored output, pool that produced the input or None)
  x = np.concatenate([q(fg[..., 3:]), ofg_q], axis=0)
  vt, pend = [], None
  for item in ref.VGG_SPEC:
    if item == 'pool':
      yv, idx = ops.maxpool2x2_fwd(x)
      pend = (x.shape, idx)
      x = yv
    else:
      yv = q(ops.relu(ops.conv2d_fwd(x, q(p['vgg_16/%s/weights' % item[0]]), f32(p['vgg_16/%s/biases' % item[0]]), 1, 1)))
      vt.append((item[0], x, yv, pend))
      pend = None
      x = yv
  fa, fb = x[:N], x[N:]
  content = ((fa - fb) ** 2).sum() / 2 / fa.size
  gen_loss_l1 = np.mean(np.abs(tgt - outputs)) + np.mean(np.abs(masks - alphas)) + content
  nodes = dict(Outputs_raw=outputs, Outputs_FG=outputs_fg, Discrim_loss=discrim_loss, Gen_loss_GAN=gen_loss_gan, Gen_loss_L1=gen_loss_l1,
               Gen_loss=gen_loss_gan * gan_weight + gen_loss_l1 * l1_weight, Perceptual_loss=content)
  # ---- D loss gradients: one pass over the 3N batch ----
  dpr = -2.0 / (predict_real + eps) / M * 0.5
  seed = np.concatenate([dpr * p0 * (1 - p0), dpr * p1 * (1 - p1), 1.0 / (1 - pf + eps) / M * pf * (1 - pf)], axis=0)
  dgr, _ = D.backward(q(seed))
  d_dy_dloss = dict(D.dy_log)
  # ---- G loss gradients ----
  seed_g = gan_weight * (-1.0 / (pf + eps)) / M * pf * (1 - pf)
  _, din = D.backward(q(seed_g), sub=(slice(2 * N, 3 * N), 2), want_dw=False)
  d_din = din['d_inputs'][..., 3:]
  # perceptual seed (through conv3_3's relu), then dX-only backward over the fake half, mirroring the device:
  # bwd-data epilogue multiplies by relu'(stored producer output) before storing; a pool stores the raw gradient
  # of its output and its backward kernel routes it to the arg-max and applies the producer's relu mask.
  d = q(np.where(fb > 0, l1_weight * (fb - fa) / fa.size, 0.0))
  d_vin = None
  for li in range(len(vt) - 1, -1, -1):
    name, xin, yv, pool = vt[li]
    g, _, _ = ops.conv2d_bwd(xin[N:], q(p['vgg_16/%s/weights' % name]), d, 1, 1, need_dx=True, need_dw=False)
    if li == 0:
      d_vin = q(g)
      break
    prev_y = vt[li - 1][2][N:]
    if pool is not None:
      shape, idx = pool
      routed = ops.maxpool2x2_bwd(q(g), idx[N:], (N,) + tuple(shape[1:]))
      d = routed * (prev_y > 0)
    else:
      d = q(g * (prev_y > 0))
  d_ofg = d_din + d_vin
  s = l1_weight / outputs.size
  d_outputs = f32(-s * np.sign(tgt - outputs))
  d_alphas = f32(-s * np.sign(masks - alphas))
  dout4 = ref.composite_bwd(out4, tgt, d_outputs, d_alphas, d_ofg)
  dy4 = q(dout4 * (1 - out4 ** 2))
  ggr, _ = G.backward(dy4)
  nodes.update(Discrim_grads=dgr, Gen_grads=ggr, G=G, D=D, d_dy_dloss=d_dy_dloss, d_vin=d_vin, d_din=d_din, dy4=dy4)
  return nodes
