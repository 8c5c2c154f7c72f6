"""Oracle: PixReferNet generator / discriminator / VGG trunk / losses / TF-Adam.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no
reference-captured vectors exist; TF1.x cannot run in the build container).

Restates, in plain numpy:
  voicepuppet/pixrefer/pixrefer.py:59-330   build_network (G, composite, 3x D, VGG)
  voicepuppet/pixrefer/pixrefer.py:332-354  add_cost_function
  voicepuppet/pixrefer/pixrefer.py:356-412  build_train_op (pre/deprocess, LR, 2x Adam)
  voicepuppet/pixrefer/pixrefer.py:414-438  build_inference_op
  voicepuppet/pixrefer/vgg_simple.py:96-162 vgg_16 up to conv3_3

Parameters live in a dict keyed by the TF variable names (SURVEY.md 8a manifest).

Documented semantic choices (the reference leaves them implementation-defined):
  * both gradients (D-loss and G-loss) are taken from ONE forward with the
    pre-update weights; then Adam(D), then Adam(G)  (SURVEY.md 3.1).
  * the bias of a conv that is followed by batch-norm has an analytically zero
    gradient; it is set to exactly 0 instead of float round-off noise.
"""
import numpy as np

from . import nn_ops as ops


# ----------------------------------------------------------------------------
# topology tables
# ----------------------------------------------------------------------------
def generator_spec(ngf=64):
  """(scope, kind, sources, cout, has_bn, pre_activation) in forward order.
  pixrefer.py:166-277."""
  L = [('encoder_1', 'conv', ['inputs'], ngf, False, None)]
  for k, c in zip((2, 3, 4), (ngf * 2, ngf * 2, ngf * 4)):
    L.append(('encoder_%d' % k, 'conv', ['encoder_%d' % (k - 1)], c, True, 'lrelu'))
  L.append(('encoder_fg_1', 'conv', ['fg_inputs'], ngf, False, None))
  for k, c in zip((2, 3, 4), (ngf * 2, ngf * 2, ngf * 4)):
    L.append(('encoder_fg_%d' % k, 'conv', ['encoder_fg_%d' % (k - 1)], c, True, 'lrelu'))
  L.append(('merged_encoder_2', 'conv', ['encoder_4', 'encoder_fg_4'], ngf * 4, True, 'lrelu'))
  for k in (3, 4, 5):
    L.append(('merged_encoder_%d' % k, 'conv', ['merged_encoder_%d' % (k - 1)], ngf * 8, True, 'lrelu'))
  L.append(('merged_decoder_5', 'deconv', ['merged_encoder_5'], ngf * 8, True, 'relu'))
  L.append(('merged_decoder_4', 'deconv', ['merged_decoder_5', 'merged_encoder_4'], ngf * 8, True, 'relu'))
  L.append(('merged_decoder_3', 'deconv', ['merged_decoder_4', 'merged_encoder_3'], ngf * 4, True, 'relu'))
  L.append(('merged_decoder_2', 'deconv', ['merged_decoder_3', 'merged_encoder_2'], ngf * 4, True, 'relu'))
  L.append(('merged2_decoder_4', 'deconv', ['merged_decoder_2', 'encoder_4'], ngf * 2, True, 'relu'))
  L.append(('merged2_decoder_3', 'deconv', ['merged2_decoder_4', 'encoder_3'], ngf * 2, True, 'relu'))
  L.append(('merged2_decoder_2', 'deconv', ['merged2_decoder_3', 'encoder_2'], ngf, True, 'relu'))
  L.append(('decoder_1', 'deconv', ['merged2_decoder_2', 'encoder_1'], 4, False, 'relu'))
  return L


def discriminator_spec(ndf=64):
  """(scope, cout, stride, has_bn).  pixrefer.py:103-134."""
  return [('layer_1', ndf, 2, False), ('layer_2', ndf * 2, 2, True), ('layer_3', ndf * 4, 2, True),
          ('layer_4', ndf * 8, 1, True), ('layer_5', 1, 1, False)]


VGG_SPEC = [('conv1/conv1_1', 3, 64), ('conv1/conv1_2', 64, 64), 'pool',
            ('conv2/conv2_1', 64, 128), ('conv2/conv2_2', 128, 128), 'pool',
            ('conv3/conv3_1', 128, 256), ('conv3/conv3_2', 256, 256), ('conv3/conv3_3', 256, 256)]


def _src_channels(name, spec_out, in_ch):
  return in_ch[name] if name in in_ch else spec_out[name]


def param_manifest(ngf=64, ndf=64):
  """Ordered [(tf_name, shape)] for the generator and the discriminator."""
  g, d = [], []
  in_ch = {'inputs': 6, 'fg_inputs': 3}
  outc = {}
  for scope, kind, srcs, cout, bn, _ in generator_spec(ngf):
    cin = sum(_src_channels(s, outc, in_ch) for s in srcs)
    outc[scope] = cout
    if kind == 'conv':
      g.append(('generator/%s/conv2d/kernel' % scope, (4, 4, cin, cout)))
      g.append(('generator/%s/conv2d/bias' % scope, (cout,)))
    else:
      g.append(('generator/%s/conv2d_transpose/kernel' % scope, (4, 4, cout, cin)))
      g.append(('generator/%s/conv2d_transpose/bias' % scope, (cout,)))
    if bn:
      g.append(('generator/%s/batch_normalization/gamma' % scope, (cout,)))
      g.append(('generator/%s/batch_normalization/beta' % scope, (cout,)))
  cin = 6
  for scope, cout, _, bn in discriminator_spec(ndf):
    d.append(('discriminator/%s/conv2d/kernel' % scope, (4, 4, cin, cout)))
    d.append(('discriminator/%s/conv2d/bias' % scope, (cout,)))
    if bn:
      d.append(('discriminator/%s/batch_normalization/gamma' % scope, (cout,)))
      d.append(('discriminator/%s/batch_normalization/beta' % scope, (cout,)))
    cin = cout
  return g, d


def vgg_manifest():
  out = []
  for item in VGG_SPEC:
    if item == 'pool':
      continue
    name, cin, cout = item
    out.append(('vgg_16/%s/weights' % name, (3, 3, cin, cout)))
    out.append(('vgg_16/%s/biases' % name, (cout,)))
  return out


def init_params(ngf=64, ndf=64, seed=0, dtype=np.float64):
  """TF initialisers of the reference: kernels N(0,0.02) (pixrefer.py:64,68),
  bias 0, gamma N(1,0.02), beta 0 (pixrefer.py:100-101).  VGG: synthetic
  He-normal stand-in (the real vgg_16.ckpt is an external download)."""
  rng = np.random.default_rng(seed)
  p = {}
  g, d = param_manifest(ngf, ndf)
  for name, shape in g + d:
    if name.endswith('kernel'):
      p[name] = rng.normal(0, 0.02, shape).astype(dtype)
    elif name.endswith('gamma'):
      p[name] = rng.normal(1.0, 0.02, shape).astype(dtype)
    else:
      p[name] = np.zeros(shape, dtype)
  for name, shape in vgg_manifest():
    if name.endswith('weights'):
      fan_in = shape[0] * shape[1] * shape[2]
      p[name] = rng.normal(0, np.sqrt(2.0 / fan_in), shape).astype(dtype)
    else:
      p[name] = rng.normal(0, 0.05, shape).astype(dtype)
  return p


_ACT = {None: (lambda x: x, lambda x: np.ones_like(x)),
        'lrelu': (lambda x: ops.lrelu(x, 0.2), lambda x: ops.lrelu_grad(x, 0.2)),
        'relu': (ops.relu, ops.relu_grad)}


# ----------------------------------------------------------------------------
# generator  (pixrefer.py:166-277)
# ----------------------------------------------------------------------------
def generator_fwd(p, inputs, fg_inputs3, ngf=64):
  """inputs [N,H,H,6], fg_inputs3 [N,H,H,3] (already in [-1,1]) -> out [N,H,H,4], tape."""
  acts = {'inputs': inputs, 'fg_inputs': fg_inputs3}
  tape = {}
  for scope, kind, srcs, cout, bn, pre in generator_spec(ngf):
    x = acts[srcs[0]] if len(srcs) == 1 else np.concatenate([acts[s] for s in srcs], axis=3)
    xa = _ACT[pre][0](x)
    if kind == 'conv':
      w = p['generator/%s/conv2d/kernel' % scope]
      b = p['generator/%s/conv2d/bias' % scope]
      y = ops.conv2d_fwd(xa, w, b, 2, 1)
    else:
      w = p['generator/%s/conv2d_transpose/kernel' % scope]
      b = p['generator/%s/conv2d_transpose/bias' % scope]
      y = ops.deconv4s2_fwd(xa, w, b)
    rec = {'x': x, 'xa': xa, 'y': y}
    if bn:
      z, rec['bn'] = ops.bn_train_fwd(y, p['generator/%s/batch_normalization/gamma' % scope],
                                      p['generator/%s/batch_normalization/beta' % scope])
    elif scope == 'decoder_1':
      z = np.tanh(y)
    else:
      z = y
    acts[scope] = z
    tape[scope] = rec
  return acts['decoder_1'], (acts, tape)


def generator_bwd(p, cache, dout, ngf=64):
  """dout = dL/d(generator output, post-tanh).  Returns ({name: grad}, dacts)."""
  acts, tape = cache
  grads = {}
  dacts = {k: np.zeros_like(v) for k, v in acts.items()}
  dacts['decoder_1'] = dacts['decoder_1'] + dout
  for scope, kind, srcs, cout, bn, pre in reversed(generator_spec(ngf)):
    rec = tape[scope]
    dz = dacts[scope]
    if bn:
      dy, dg, db_ = ops.bn_train_bwd(dz, rec['bn'])
      grads['generator/%s/batch_normalization/gamma' % scope] = dg
      grads['generator/%s/batch_normalization/beta' % scope] = db_
    elif scope == 'decoder_1':
      dy = dz * (1.0 - acts[scope] ** 2)
    else:
      dy = dz
    if kind == 'conv':
      w = p['generator/%s/conv2d/kernel' % scope]
      dxa, dw, dbias = ops.conv2d_bwd(rec['xa'], w, dy, 2, 1)
      grads['generator/%s/conv2d/kernel' % scope] = dw
      grads['generator/%s/conv2d/bias' % scope] = np.zeros_like(dbias) if bn else dbias
    else:
      w = p['generator/%s/conv2d_transpose/kernel' % scope]
      dxa, dw, dbias = ops.deconv4s2_bwd(rec['xa'], w, dy)
      grads['generator/%s/conv2d_transpose/kernel' % scope] = dw
      grads['generator/%s/conv2d_transpose/bias' % scope] = np.zeros_like(dbias) if bn else dbias
    dx = dxa * _ACT[pre][1](rec['x'])
    c0 = 0
    for s in srcs:
      c = acts[s].shape[3]
      dacts[s] = dacts[s] + dx[..., c0:c0 + c]
      c0 += c
  return grads, dacts


def composite(out4, targets):
  """pixrefer.py:281-286.  Returns Outputs, Alphas(tiled x3), Outputs_FG."""
  rgb = out4[..., :3]
  alpha = np.tile((out4[..., 3:] + 1) / 2, (1, 1, 1, 3))
  return rgb * alpha + targets * (1 - alpha), alpha, rgb * alpha + alpha - 1


def composite_bwd(out4, targets, d_outputs, d_alphas, d_outputs_fg):
  rgb = out4[..., :3]
  alpha = (out4[..., 3:] + 1) / 2
  dout = np.zeros_like(out4)
  dout[..., :3] = (d_outputs + d_outputs_fg) * alpha
  dalpha = (d_outputs * (rgb - targets)).sum(axis=3, keepdims=True) \
      + (d_outputs_fg * (rgb + 1)).sum(axis=3, keepdims=True) + d_alphas.sum(axis=3, keepdims=True)
  dout[..., 3:] = dalpha / 2
  return dout


# ----------------------------------------------------------------------------
# discriminator  (pixrefer.py:103-134); one application = its own batch stats
# ----------------------------------------------------------------------------
def discriminator_fwd(p, cond3, img3, ndf=64, f32_probs=False):
  """f32_probs: the sigmoid output is rounded to float32, as the reference's float32 graph holds it (see forward_backward)."""
  x = np.concatenate([cond3, img3], axis=3)
  tape = []
  for scope, cout, stride, bn in discriminator_spec(ndf):
    w = p['discriminator/%s/conv2d/kernel' % scope]
    b = p['discriminator/%s/conv2d/bias' % scope]
    y = ops.conv2d_fwd(x, w, b, stride, 1)
    rec = {'x': x, 'y': y}
    if bn:
      z, rec['bn'] = ops.bn_train_fwd(y, p['discriminator/%s/batch_normalization/gamma' % scope],
                                      p['discriminator/%s/batch_normalization/beta' % scope])
    else:
      z = y
    rec['z'] = z
    x = ops.sigmoid(z) if scope == 'layer_5' else ops.lrelu(z, 0.2)
    if scope == 'layer_5' and f32_probs:
      x = x.astype(np.float32)
      rec['f32'] = True
    rec['out'] = x
    tape.append(rec)
  return x, tape


def discriminator_bwd(p, tape, dp, ndf=64, need_dw=True):
  """dp = dL/d(sigmoid output).  Returns (grads, d(concat input) [N,H,H,6])."""
  grads = {}
  dout = dp
  spec = discriminator_spec(ndf)
  for li in range(len(spec) - 1, -1, -1):
    scope, cout, stride, bn = spec[li]
    rec = tape[li]
    if scope == 'layer_5' and rec.get('f32'):
      o = rec['out']                                    # float32: 1 - o is exactly 0 once the sigmoid has saturated
      dz = (dout.astype(np.float32) * o * (np.float32(1) - o)).astype(np.float64)
    elif scope == 'layer_5':
      dz = dout * rec['out'] * (1 - rec['out'])
    else:
      dz = dout * ops.lrelu_grad(rec['z'], 0.2)
    if bn:
      dy, dg, db_ = ops.bn_train_bwd(dz, rec['bn'])
      grads['discriminator/%s/batch_normalization/gamma' % scope] = dg
      grads['discriminator/%s/batch_normalization/beta' % scope] = db_
    else:
      dy = dz
    w = p['discriminator/%s/conv2d/kernel' % scope]
    dx, dw, dbias = ops.conv2d_bwd(rec['x'], w, dy, stride, 1, need_dx=True, need_dw=need_dw)
    if need_dw:
      grads['discriminator/%s/conv2d/kernel' % scope] = dw
      grads['discriminator/%s/conv2d/bias' % scope] = np.zeros_like(dbias) if bn else dbias
    dout = dx
  return grads, dout


# ----------------------------------------------------------------------------
# VGG-16 trunk to conv3_3 (vgg_simple.py:138-151): conv3x3 s1 SAME + bias + relu
# ----------------------------------------------------------------------------
def vgg_fwd(p, x):
  tape = []
  for item in VGG_SPEC:
    if item == 'pool':
      y, idx = ops.maxpool2x2_fwd(x)
      tape.append(('pool', x.shape, idx))
      x = y
    else:
      name = item[0]
      y = ops.relu(ops.conv2d_fwd(x, p['vgg_16/%s/weights' % name], p['vgg_16/%s/biases' % name], 1, 1))
      tape.append(('conv', name, x, y))
      x = y
  return x, tape


def vgg_bwd(p, tape, df3):
  """dX only: the VGG weights are frozen (not in gen_tvars, pixrefer.py:404)."""
  d = df3
  for rec in reversed(tape):
    if rec[0] == 'pool':
      d = ops.maxpool2x2_bwd(d, rec[2], rec[1])
    else:
      _, name, x, y = rec
      d = d * (y > 0)
      d, _, _ = ops.conv2d_bwd(x, p['vgg_16/%s/weights' % name], d, 1, 1, need_dx=True, need_dw=False)
  return d


# ----------------------------------------------------------------------------
# TF Adam (tf.train.AdamOptimizer): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
#   theta -= lr_t * m / (sqrt(v) + eps)     (eps OUTSIDE the bias correction)
# ----------------------------------------------------------------------------
def adam_tf(param, grad, m, v, t, lr, beta1=0.5, beta2=0.999, eps=1e-8):
  lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
  m[...] = beta1 * m + (1 - beta1) * grad
  v[...] = beta2 * v + (1 - beta2) * grad * grad
  param -= lr_t * m / (np.sqrt(v) + eps)


def learning_rate(base_lr, global_step, decay_steps, decay_rate):
  """tf.train.exponential_decay(staircase=True), pixrefer.py:391-392."""
  return base_lr * decay_rate ** (global_step // decay_steps)


# ----------------------------------------------------------------------------
# the whole training graph: forward, losses, both gradients  (build_train_op)
# ----------------------------------------------------------------------------
def forward_backward(p, inputs, fg_inputs, targets, masks, ngf=64, ndf=64,
                     l1_weight=500.0, gan_weight=1.0, want_grads=True, f32_probs=False):
  """inputs/fg_inputs [N,H,H,6], targets/masks [N,H,H,3], all in [0,1] as the
  data generator yields them (generator.py:1011-1019).  Returns a dict with the
  `nodes` of pixrefer.py:356-412 plus gradient dicts.

  f32_probs: evaluate the GAN terms the way the reference's float32 graph does (pixrefer.py:336-345: tf.log(1 - predict_fake
  + EPS) on a float32 sigmoid output).  Once the discriminator saturates (predict_fake within 6e-8 of 1, which happens after
  ONE Adam step at the reference's learning rate) float32 holds predict_fake == 1 exactly: the loss term is -log(1e-12) and
  its gradient through the sigmoid is exactly 0, where exact arithmetic gives -log(1 - p) and a gradient of 1.  This is a
  property of the reference's dtype, not of an implementation, so multi-step trajectories are compared with it switched on."""
  inp = inputs * 2 - 1          # preprocess, pixrefer.py:373-375 (masks are NOT preprocessed)
  fg = fg_inputs * 2 - 1
  tgt = targets * 2 - 1
  nodes = {}

  out4, gcache = generator_fwd(p, inp, fg[..., :3], ngf)
  outputs, alphas, outputs_fg = composite(out4, tgt)

  # three discriminator applications, shared weights, separate batch statistics (pixrefer.py:295-306)
  p_real1, t_real1 = discriminator_fwd(p, inp[..., 3:], fg[..., 3:], ndf, f32_probs)
  p_real2, t_real2 = discriminator_fwd(p, inp[..., :3], fg[..., :3], ndf, f32_probs)
  predict_real = (p_real1 + p_real2) * (np.float32(0.5) if f32_probs else 0.5)
  predict_fake, t_fake = discriminator_fwd(p, inp[..., 3:], outputs_fg, ndf, f32_probs)

  # perceptual loss (pixrefer.py:318-323)
  n = inp.shape[0]
  f3, vtape = vgg_fwd(p, np.concatenate([fg[..., 3:], outputs_fg], axis=0))
  fa, fb = f3[:n], f3[n:]
  content_loss = ((fa - fb) ** 2).sum() / 2 / fa.size

  eps = np.float32(1e-12) if f32_probs else 1e-12
  one = np.float32(1) if f32_probs else 1.0
  discrim_loss = np.mean(-(np.log(predict_real + eps) * 2 + np.log(one - predict_fake + eps)), dtype=np.float64)
  gen_loss_gan = np.mean(-np.log(predict_fake + eps), dtype=np.float64)
  gen_loss_l1 = np.mean(np.abs(tgt - outputs)) + np.mean(np.abs(masks - alphas)) + content_loss
  gen_loss = gen_loss_gan * gan_weight + gen_loss_l1 * l1_weight

  nodes.update(Outputs=(outputs + 1) / 2, Outputs_raw=outputs, Alphas=alphas, Outputs_FG=outputs_fg,
               Predict_real=predict_real, Predict_fake=predict_fake, Perceptual_loss=content_loss,
               Discrim_loss=discrim_loss, Gen_loss_GAN=gen_loss_gan, Gen_loss_L1=gen_loss_l1,
               Gen_loss=gen_loss, gen_out4=out4, g_acts=gcache[0])
  if not want_grads:
    return nodes

  # ---- discriminator gradients (var_list = discriminator*) ----
  m = predict_real.size
  d_preal = (-2 * one) / (predict_real + eps) / m
  d_pfake = one / (one - predict_fake + eps) / m
  dgr = {}
  for tape, dp in ((t_real1, d_preal / 2), (t_real2, d_preal / 2), (t_fake, d_pfake)):
    g, _ = discriminator_bwd(p, tape, dp, ndf)
    for k, v in g.items():
      dgr[k] = dgr.get(k, 0) + v

  # ---- generator gradients (var_list = generator*) ----
  d_pfake_g = gan_weight * (-one / (predict_fake + eps) / m)
  _, d_dinput = discriminator_bwd(p, t_fake, d_pfake_g, ndf, need_dw=False)
  d_outputs_fg = d_dinput[..., 3:].copy()
  df3 = np.zeros_like(f3)
  df3[n:] = l1_weight * (fb - fa) / fa.size
  d_vin = vgg_bwd(p, [(_slice_rec(r, n)) for r in vtape], df3[n:])
  d_outputs_fg += d_vin
  d_outputs = l1_weight * (-np.sign(tgt - outputs)) / outputs.size
  d_alphas = l1_weight * (-np.sign(masks - alphas)) / alphas.size
  dout4 = composite_bwd(out4, tgt, d_outputs, d_alphas, d_outputs_fg)
  ggr, g_dacts = generator_bwd(p, gcache, dout4, ngf)
  nodes.update(Discrim_grads=dgr, Gen_grads=ggr, d_gen_out4=dout4, d_outputs_fg=d_outputs_fg, g_dacts=g_dacts,
               d_dinput=d_dinput, d_vin=d_vin)
  return nodes


def _slice_rec(rec, n):
  """Restrict a VGG tape record to the generated half of the 2N batch."""
  if rec[0] == 'pool':
    shape = (rec[1][0] - n,) + tuple(rec[1][1:])
    return ('pool', shape, rec[2][n:])
  return ('conv', rec[1], rec[2][n:], rec[3][n:])


class TrainState:
  """Parameters + the two Adam states + global_step (pixrefer.py:390-407)."""

  def __init__(self, params, ngf=64, ndf=64, base_lr=3e-4, beta1=0.5, decay_steps=1000, decay_rate=0.999, f32_probs=False):
    self.p = params
    self.f32_probs = f32_probs
    self.ngf, self.ndf = ngf, ndf
    self.base_lr, self.beta1 = base_lr, beta1
    self.decay_steps, self.decay_rate = decay_steps, decay_rate
    self.global_step = 0
    self.t_d = 0
    self.t_g = 0
    g, d = param_manifest(ngf, ndf)
    self.g_names = [n for n, _ in g]
    self.d_names = [n for n, _ in d]
    self.m = {n: np.zeros_like(params[n]) for n in self.g_names + self.d_names}
    self.v = {n: np.zeros_like(params[n]) for n in self.g_names + self.d_names}

  def step(self, inputs, fg_inputs, targets, masks, l1_weight=500.0, gan_weight=1.0):
    nodes = forward_backward(self.p, inputs, fg_inputs, targets, masks, self.ngf, self.ndf, l1_weight, gan_weight,
                             f32_probs=self.f32_probs)
    lr = learning_rate(self.base_lr, self.global_step, self.decay_steps, self.decay_rate)
    self.t_d += 1
    for nme in self.d_names:
      adam_tf(self.p[nme], nodes['Discrim_grads'][nme], self.m[nme], self.v[nme], self.t_d, lr, self.beta1)
    self.global_step += 1
    self.t_g += 1
    for nme in self.g_names:
      adam_tf(self.p[nme], nodes['Gen_grads'][nme], self.m[nme], self.v[nme], self.t_g, lr, self.beta1)
    self.global_step += 1
    nodes['Lr'] = lr
    nodes['Global_step'] = self.global_step
    return nodes


def inference(p, inputs, fg_inputs3, targets, ngf=64):
  """build_inference_op, pixrefer.py:414-438 (inputs in [0,1])."""
  inp, fg, tgt = inputs * 2 - 1, fg_inputs3 * 2 - 1, targets * 2 - 1
  out4, _ = generator_fwd(p, inp, fg[..., :3], ngf)
  outputs, alphas, outputs_fg = composite(out4, tgt)
  return {'Outputs': (outputs + 1) / 2, 'Alphas': alphas,
          'Outputs_FG': ((outputs_fg + alphas - 1) + 1) / 2}
