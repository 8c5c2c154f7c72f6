"""Oracle, second opinion: the PixReferNet G+D training graph restated on torch-CPU autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED, like the numpy oracle it cross-checks.

Two users:
  * tests/test_oracle_pixrefer.py runs it in float64 against oracle/pixrefer_ref.py (independent derivation of every
    gradient: autograd here, hand-written backward passes there);
  * bench.py's `cpu_baseline` leg times it in float32 with all host cores as the "CPU restatement (TF-CPU proxy)" of
    SURVEY.md 8d: TF-CPU and torch-CPU both dispatch conv / conv-transpose / batch-norm to oneDNN-class kernels, and
    TensorFlow exists on neither box.

Follows voicepuppet/pixrefer/pixrefer.py:59-330 (network), :332-354 (losses), :356-412 (two Adam optimisers, one
forward with the pre-update weights, D first) and voicepuppet/pixrefer/vgg_simple.py:96-162.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import pixrefer_ref as ref


def _lrelu(x):
  return 0.6 * x + 0.4 * x.abs()     # tf.nn.leaky_relu(alpha=0.2) as pixrefer.py:90-97 writes it


class TorchGraph:
  """Parameters as torch leaves keyed by the TF variable names; `losses_and_grads` = one forward + both gradient sets."""

  def __init__(self, params, ngf=64, ndf=64, dtype=torch.float64, l1_weight=500.0, gan_weight=1.0):
    self.ngf, self.ndf, self.dtype = ngf, ndf, dtype
    self.l1_weight, self.gan_weight = l1_weight, gan_weight
    self.tp = {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_() for k, v in params.items()}
    gn, dn = ref.param_manifest(ngf, ndf)
    self.g_names = [k for k, _ in gn]
    self.d_names = [k for k, _ in dn]
    self.m = {k: torch.zeros_like(self.tp[k]) for k in self.g_names + self.d_names}
    self.v = {k: torch.zeros_like(self.tp[k]) for k in self.g_names + self.d_names}
    self.t_d = self.t_g = 0
    self.global_step = 0

  def _nchw(self, a):
    return torch.as_tensor(np.asarray(a), dtype=self.dtype).permute(0, 3, 1, 2)

  def forward(self, inputs, fg_inputs, targets, masks):
    tp = self.tp
    inp, fg, tgt = self._nchw(inputs) * 2 - 1, self._nchw(fg_inputs) * 2 - 1, self._nchw(targets) * 2 - 1
    msk = self._nchw(masks)
    acts = {'inputs': inp, 'fg_inputs': fg[:, :3]}
    for scope, kind, srcs, cout, bn, pre in ref.generator_spec(self.ngf):
      x = torch.cat([acts[s] for s in srcs], 1)
      x = {None: lambda v: v, 'lrelu': _lrelu, 'relu': torch.relu}[pre](x)
      if kind == 'conv':
        y = F.conv2d(x, tp['generator/%s/conv2d/kernel' % scope].permute(3, 2, 0, 1).contiguous(),
                     tp['generator/%s/conv2d/bias' % scope], 2, 1)
      else:
        y = F.conv_transpose2d(x, tp['generator/%s/conv2d_transpose/kernel' % scope].permute(3, 2, 0, 1).contiguous(),
                               tp['generator/%s/conv2d_transpose/bias' % scope], 2, 1)
      if bn:
        y = F.batch_norm(y, None, None, tp['generator/%s/batch_normalization/gamma' % scope],
                         tp['generator/%s/batch_normalization/beta' % scope], True, 0.1, 1e-5)
      acts[scope] = y
    out = torch.tanh(acts['decoder_1'])
    rgb, alpha = out[:, :3], ((out[:, 3:] + 1) / 2).repeat(1, 3, 1, 1)
    outputs = rgb * alpha + tgt * (1 - alpha)
    outputs_fg = rgb * alpha + alpha - 1

    def disc(a, b):
      x = torch.cat([a, b], 1)
      for scope, cout, stride, bn in ref.discriminator_spec(self.ndf):
        x = F.conv2d(x, tp['discriminator/%s/conv2d/kernel' % scope].permute(3, 2, 0, 1).contiguous(),
                     tp['discriminator/%s/conv2d/bias' % scope], stride, 1)
        if bn:
          x = F.batch_norm(x, None, None, tp['discriminator/%s/batch_normalization/gamma' % scope],
                           tp['discriminator/%s/batch_normalization/beta' % scope], True, 0.1, 1e-5)
        x = torch.sigmoid(x) if scope == 'layer_5' else _lrelu(x)
      return x

    p_real = (disc(inp[:, 3:], fg[:, 3:]) + disc(inp[:, :3], fg[:, :3])) / 2
    p_fake = disc(inp[:, 3:], outputs_fg)
    x = torch.cat([fg[:, 3:], outputs_fg], 0)
    for item in ref.VGG_SPEC:
      if item == 'pool':
        x = F.max_pool2d(x, 2)
      else:
        x = torch.relu(F.conv2d(x, tp['vgg_16/%s/weights' % item[0]].permute(3, 2, 0, 1).contiguous(),
                                tp['vgg_16/%s/biases' % item[0]], 1, 1))
    n = inp.shape[0]
    content = ((x[:n] - x[n:]) ** 2).sum() / 2 / x[:n].numel()
    d_loss = (-(torch.log(p_real + 1e-12) * 2 + torch.log(1 - p_fake + 1e-12))).mean()
    g_gan = (-torch.log(p_fake + 1e-12)).mean()
    g_l1 = (tgt - outputs).abs().mean() + (msk - alpha).abs().mean() + content
    g_loss = g_gan * self.gan_weight + g_l1 * self.l1_weight
    return dict(d_loss=d_loss, g_gan=g_gan, g_l1=g_l1, content=content, g_loss=g_loss, outputs=outputs,
                outputs_fg=outputs_fg, alpha=alpha, p_real=p_real, p_fake=p_fake)

  def losses_and_grads(self, inputs, fg_inputs, targets, masks):
    f = self.forward(inputs, fg_inputs, targets, masks)
    dgr = torch.autograd.grad(f['d_loss'], [self.tp[k] for k in self.d_names], retain_graph=True)
    ggr = torch.autograd.grad(f['g_loss'], [self.tp[k] for k in self.g_names])
    return f, dict(zip(self.d_names, dgr)), dict(zip(self.g_names, ggr))

  def step(self, inputs, fg_inputs, targets, masks, base_lr=3e-4, beta1=0.5, beta2=0.999, eps=1e-8,
           decay_steps=1000, decay_rate=0.999):
    """train_pixrefer.py:136-143: both gradients from one forward, TF-Adam on discriminator* then generator*."""
    f, dgr, ggr = self.losses_and_grads(inputs, fg_inputs, targets, masks)
    lr = ref.learning_rate(base_lr, self.global_step, decay_steps, decay_rate)
    self.t_d += 1
    self.t_g += 1
    with torch.no_grad():
      for names, grads, t in ((self.d_names, dgr, self.t_d), (self.g_names, ggr, self.t_g)):
        lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
        for k in names:
          g = grads[k]
          self.m[k].mul_(beta1).add_(g, alpha=1 - beta1)
          self.v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
          self.tp[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))
        self.global_step += 1
    return {k: float(f[k].detach()) for k in ('d_loss', 'g_gan', 'g_l1', 'content', 'g_loss')}


def torch_graph(p, inputs, fg_inputs, targets, masks, ngf, ndf):
  """float64 losses, composite output and every gradient as numpy (the form tests/test_oracle_pixrefer.py compares)."""
  g = TorchGraph(p, ngf, ndf, torch.float64)
  f, dgr, ggr = g.losses_and_grads(inputs, fg_inputs, targets, masks)
  return dict(d_loss=f['d_loss'].item(), g_gan=f['g_gan'].item(), g_l1=f['g_l1'].item(), content=f['content'].item(),
              outputs=f['outputs'].detach().permute(0, 2, 3, 1).numpy(),
              dgr={k: v.numpy() for k, v in dgr.items()}, ggr={k: v.numpy() for k, v in ggr.items()})
