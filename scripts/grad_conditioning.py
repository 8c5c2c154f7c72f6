#!/usr/bin/env python3
"""How well-conditioned are the generator's gradients at the N = 4 full-width fixture?  (CPU only, ~1 minute)

Runs the generator of the rounding-aware oracle (oracle/pixrefer_lowp_ref.py dataflow) forward and backward on
tests/golden/full_width_n4.npz (or on four unrelated smooth random images) with a fixed random output gradient, under different
rounding policies, and prints the element-wise rel-L2 distance of every gradient tensor to the all-float64 run:
  bf16        weights, stored activations and stored gradients rounded to bf16 (what any bf16 implementation does)
  hi          the same with the seven few-pixel batch-normalised tensors (and their accumulated gradients) in float32 (round 4)
  w_only / a_only / g_only   ONLY the packed weights / ONLY the stored activations / ONLY the stored gradients rounded
  exact_bot / exact_deep     bf16 everywhere except float32 COMPUTE (weights, operands, outputs) in the eight bottleneck layers / in
                             the bottleneck plus the 16x16 and 32x32 layers around it
usage: python scripts/grad_conditioning.py fixture|random bf16 hi w_only a_only g_only exact_bot exact_deep
Measured (profiles/r04_grad_conditioning.txt): weights alone 0.25-0.28, activations alone 0.34, gradients alone 0.009-0.015."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pixrefer_lowp_ref as lowp, pixrefer_ref as ref, nn_ops as ops
from oracle.pixrefer_lowp_ref import f32, _bn_affine, _per_group, ACT, ACT_GRAD
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'full_width_n4.npz'))
ngf = int(d['ngf'])
p = ref.init_params(ngf, ngf, seed=int(d['seed']), dtype=np.float32)
p = {k: v.astype(np.float64) for k, v in p.items() if k.startswith('generator')}
def load(kind):
  if kind == 'fixture':
    inp = d['inputs'].astype(np.float32) / 255.0; fg = d['fg_inputs'].astype(np.float32) / 255.0
  else:   # four unrelated smooth random images
    rng = np.random.default_rng(5)
    def img(c):
      x = rng.uniform(size=(4, 256, 256, c))
      for _ in range(3):
        x = (x + np.roll(x, 1, 1) + np.roll(x, -1, 1) + np.roll(x, 1, 2) + np.roll(x, -1, 2)) / 5
      x = (x - x.min()) / (x.max() - x.min())
      return x.astype(np.float32)
    inp, fg = img(6), img(6)
  return f32(f32(inp) * 2 - 1), f32(f32(fg) * 2 - 1)
rng = np.random.default_rng(0)
dy4 = rng.normal(size=(4, 256, 256, 4)) * 1e-3
BOT = ['merged_encoder_2','merged_encoder_3','merged_encoder_4','merged_encoder_5','merged_decoder_5','merged_decoder_4','merged_decoder_3']

class Net3(lowp.Net):
  """q split into qw (packed weights), qa (stored activations / outputs), qg (stored gradients); `exact` = layers whose weights, input
  activations and outputs are kept unrounded (float32 compute)."""
  def __init__(self, spec, params, qw, qa, qg, hi=(), exact=()):
    lowp.Net.__init__(self, spec, params, 'generator', qa, 1, hi)
    self.qw, self.qa, self.qg, self.exact = qw, qa, qg, frozenset(exact)
  def forward(self, inputs):
    self.y.update(inputs)
    self.xa_exact = {}
    need = {}
    for scope, kind, srcs, cout, bn, pre, stride, final in self.spec:
      for s in srcs:
        if pre is not None: need.setdefault(s, set()).add(pre)
    for scope, kind, srcs, cout, bn, pre, stride, final in self.spec:
      ex = scope in self.exact
      xs = [self.y[s] if pre is None else (self.xa_exact if ex else self.xa)[(s, pre)] for s in srcs]
      x = xs[0] if len(xs) == 1 else np.concatenate(xs, axis=3)
      w = (f32 if ex else self.qw)(self.p[self.wname(scope, kind)])
      bias = None if bn else f32(self.p[self.wname(scope, kind).replace('kernel', 'bias')])
      y = ops.conv2d_fwd(x, w, bias, stride, 1) if kind == 'conv' else ops.deconv4s2_fwd(x, w, bias)
      y = f32(y) if (final or scope in self.hi or ex) else self.qa(y)
      self.y[scope] = y
      if bn:
        g = f32(self.p['%s/%s/batch_normalization/gamma' % (self.prefix, scope)]); b = f32(self.p['%s/%s/batch_normalization/beta' % (self.prefix, scope)])
        self.bn[scope] = _bn_affine(y, g, b, 1) + (g,)
      for a in need.get(scope, ()):
        z = y
        if bn:
          sc, sh = self.bn[scope][0], self.bn[scope][1]
          z = _per_group(y, sc, 1) * y + _per_group(y, sh, 1)
        self.xa[(scope, a)] = self.qa(ACT[a](z))
        self.xa_exact[(scope, a)] = f32(ACT[a](z))
    return self.y[self.spec[-1][0]]
  def backward(self, dy_last):
    dz = {self.spec[-1][0]: dy_last}
    grads = {}
    for scope, kind, srcs, cout, bn, pre, stride, final in reversed(self.spec):
      ex = scope in self.exact
      dd = dz[scope]
      if bn:
        sc, sh, mu, rs, gamma = self.bn[scope]
        y = self.y[scope]
        zh = (y - _per_group(y, mu, 1)) * _per_group(y, rs, 1)
        c1 = dd.mean(axis=(0, 1, 2))[None]; c2 = (dd * zh).mean(axis=(0, 1, 2))[None]
        grads['%s/%s/batch_normalization/gamma' % (self.prefix, scope)] = (dd * zh).sum(axis=(0, 1, 2))
        grads['%s/%s/batch_normalization/beta' % (self.prefix, scope)] = dd.sum(axis=(0, 1, 2))
        dd = (f32 if ex else self.qg)(gamma * _per_group(y, rs, 1) * (dd - _per_group(y, f32(c1), 1) - zh * _per_group(y, f32(c2), 1)))
      xs = [(self.y[s] if pre is None else (self.xa_exact if ex else self.xa)[(s, pre)]) for s in srcs]
      x = xs[0] if len(xs) == 1 else np.concatenate(xs, axis=3)
      wn = self.wname(scope, kind)
      w = (f32 if ex else self.qw)(self.p[wn])
      if kind == 'conv': dx, dw, db = ops.conv2d_bwd(x, w, dd, stride, 1, need_dx=True, need_dw=True)
      else: dx, dw, db = ops.deconv4s2_bwd(x, w, dd)
      grads[wn] = dw
      c0 = 0
      for s, xv in zip(srcs, xs):
        c = xv.shape[3]; g = dx[..., c0:c0 + c]; c0 += c
        if pre is None: continue
        g = g * ACT_GRAD[pre](xv)
        qacc = f32 if s in self.hi else self.qg
        dz[s] = qacc(g) if s not in dz else qacc(dz[s] + g)
    return grads

def run(data, qw, qa, qg, hi=(), exact=()):
  inp, fg = load(data)
  G = Net3(lowp._gspec(ngf), p, qw, qa, qg, hi, exact)
  G.forward({'inputs': qa(inp), 'fg_inputs': qa(fg[..., :3])})
  return G.backward(qg(dy4))
R, I = lowp.round_bf16, lowp.IDENT
cfgs = {'base': (I, I, I, (), ()), 'bf16': (R, R, R, (), ()), 'hi': (R, R, R, BOT, ()), 'w_only': (R, I, I, (), ()), 'a_only': (I, R, I, (), ()),
        'g_only': (I, I, R, (), ()), 'exact_bot': (R, R, R, BOT, BOT + ['merged_decoder_2']),
        'exact_deep': (R, R, R, BOT, BOT + ['merged_decoder_2', 'encoder_4', 'encoder_fg_4', 'encoder_3', 'encoder_fg_3', 'merged2_decoder_4'])}
data = sys.argv[1]
base = run(data, *cfgs['base'])
for name in sys.argv[2:]:
  t0 = time.time()
  g = run(data, *cfgs[name])
  rows = sorted(((np.linalg.norm(v - base[k]) / np.linalg.norm(base[k]), k.replace('generator/', '')) for k, v in g.items() if np.any(base[k] != 0)), reverse=True)
  deep = [a for a, k in rows if any(b + '/' in k for b in BOT + ['merged_decoder_2'])]
  rest = [a for a, k in rows if not any(b + '/' in k for b in BOT + ['merged_decoder_2'])]
  print('%s %-10s deep worst %.3f rest worst %.3f median %.3f  top: %s  (%.0fs)' % (data, name, max(deep), max(rest), np.median([a for a, _ in rows]),
        [(round(a, 3), k) for a, k in rows[:3]], time.time() - t0), flush=True)
