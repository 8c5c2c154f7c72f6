"""TEST INFRASTRUCTURE ONLY (never imported by the product path): float64 torch-CPU restatement of BFMNet.build_train_op
(voicepuppet/bfmnet/bfmnet.py:215-323 over tinynet.py:7-212), SURVEY.md 8f-4.  PARITY UNPINNED: TensorFlow 1.x cannot be imported
here and the reference ships no vectors for this path; the forward half is cross-checked against oracle/audio_ref.py (the numpy
inference restatement) in tests/test_oracle_bfmnet_train.py, the backward half is torch autograd.

What one step does, as the reference builds it:
  forward, is_training=True: every conv -> tf.contrib batch_norm with BATCH statistics (no gamma, eps 1e-3, decay 0.999,
    updates_collections=None: the moving averages are updated inside the forward; the fused kernel feeds the UNBIASED batch variance
    into the moving variance) -> relu / relu6; dropout after the encoder dense layer (tf.layers.dropout), on the GRU outputs
    (DropoutWrapper) and twice in BFMCoeffDecoder (tf.nn.dropout) - here every dropout is an explicit mask argument (entries 0 or
    1 / keep_prob) so that oracle and device see the same draw;
  add_cost_function (:215-262): vertex-space L1 on face_shape(bfm_coeffs) - face_shape([id | predicted expression]) with the
    mouth vertices weighted 10x, masked by sequence length, plus the same on first differences along time, plus
    tf.losses.get_regularization_loss() (l2_regularizer(1e-4) = 1e-4 * sum(w^2) / 2 on every conv / depthwise kernel);
  build_train_op (:291-323): AdamOptimizer(lr) (beta1 0.9, beta2 0.999, eps 1e-8) on gradients clipped by global norm."""
import numpy as np
import torch
import torch.nn.functional as F

from . import audio_ref as ar

BN_EPS, BN_DECAY, L2_SCALE = 1e-3, 0.999, 1e-4


def _same(x, k, s, value=0.0):
  """NCHW tensor padded like TF 'SAME' for kernel k = (kh, kw), stride s."""
  pt, pb, _ = ar.same_pads(x.shape[2], k[0], s[0])
  pl, pr, _ = ar.same_pads(x.shape[3], k[1], s[1])
  return F.pad(x, (pl, pr, pt, pb), value=value)


def _conv(x, w_hwio, stride=(1, 1)):
  w = w_hwio.permute(3, 2, 0, 1)
  return F.conv2d(_same(x, w.shape[2:], stride), w, stride=stride)


def _dw(x, w_hwc1):
  c = x.shape[1]
  w = w_hwc1.permute(2, 3, 0, 1)                                  # [C,1,kh,kw]
  return F.conv2d(_same(x, w.shape[2:], (1, 1)), w, groups=c)


def _pool(x, k, s):
  return F.max_pool2d(_same(x, k, s, value=float("-inf")), k, s)


class Stats:
  """batch statistics of every batch-norm of one forward: scope -> (mean, biased var, n)"""

  def __init__(self):
    self.d = {}


def _bn(x, beta, scope, stats):
  mean = x.mean(dim=(0, 2, 3))
  var = x.var(dim=(0, 2, 3), unbiased=False)
  stats.d[scope] = (mean.detach(), var.detach(), x.numel() // x.shape[1])
  return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS) + beta[None, :, None, None]


def mfccnet_train(p, x, stats, prefix="mfcc_encoder/MfccNet/"):
  """x [B,1,T5,80] (NCHW) -> [B,256,T5,3]; p: dict of torch tensors in the TF layouts."""
  s = prefix + "block0_0/conv2d"
  net = torch.relu(_bn(_conv(x, p[s + "/conv2d/kernel"], (1, 2)), p[s + "/BatchNorm/beta"], s, stats))
  for scope, cout, exp, pool in ar.MFCCNET_BLOCKS:
    b = prefix + scope
    inp = net
    net = F.relu6(_bn(_conv(net, p[b + "/expansion_1x1_conv2d/conv2d/kernel"]), p[b + "/expansion_1x1_conv2d/BatchNorm/beta"], b + "/expansion_1x1_conv2d", stats))
    net = F.relu6(_bn(_dw(net, p[b + "/depthwise_conv2d/SeparableConv2d/depthwise_weights"]), p[b + "/depthwise_conv2d/BatchNorm/beta"], b + "/depthwise_conv2d", stats))
    net = _bn(_conv(net, p[b + "/projection_1x1_conv2d/conv2d/kernel"]), p[b + "/projection_1x1_conv2d/BatchNorm/beta"], b + "/projection_1x1_conv2d", stats)
    if net.shape[1] != inp.shape[1]:
      inp = _bn(_conv(inp, p[b + "/1x1_conv2d/conv2d/kernel"]), p[b + "/1x1_conv2d/BatchNorm/beta"], b + "/1x1_conv2d", stats)
    net = net + inp
    if pool:
      net = _pool(net, (2, 2), (1, 2))
  s = prefix + "block8_0/conv2d"
  return torch.relu(_bn(_conv(net, p[s + "/conv2d/kernel"]), p[s + "/BatchNorm/beta"], s, stats))


def _lrelu(x):
  return torch.where(x >= 0, x, 0.2 * x)


def gru_seq(x, seq_len, wg, bg, wc, bc, out_mask=None):
  """tf.contrib.rnn.GRUCell under dynamic_rnn; out_mask [B,T,H]: DropoutWrapper(output_keep_prob) on the per-step OUTPUT only."""
  B, T, _ = x.shape
  H = wc.shape[1]
  h = torch.zeros(B, H, dtype=x.dtype)
  outs = []
  sl = torch.as_tensor(np.asarray(seq_len))
  for t in range(T):
    g = torch.sigmoid(torch.cat([x[:, t], h], 1) @ wg + bg)
    r, u = g[:, :H], g[:, H:]
    c = torch.tanh(torch.cat([x[:, t], r * h], 1) @ wc + bc)
    hn = u * h + (1 - u) * c
    live = (t < sl)[:, None]
    h = torch.where(live, hn, h)
    o = torch.where(live, hn, torch.zeros_like(hn))
    outs.append(o if out_mask is None else o * out_mask[:, t])
  return torch.stack(outs, 1)


def forward_train(p, ears, mfccs, seq_len, masks, stats):
  """masks: dict with 'enc' [B,T,256], 'rnn' [B,T,256], 'd0' [B,T,128], 'd1' [B,T,64] (entries 0 or 1/keep_prob; None = no dropout)."""
  B = mfccs.shape[0]
  feat = mfccnet_train(p, mfccs[:, None, :, :], stats)
  enc = _pool(feat, (5, 3), (5, 3))                                                  # [B,256,T,1]
  enc = enc[:, :, :, 0].permute(0, 2, 1)                                             # [B,T,256]
  enc = _lrelu(enc @ p["mfcc_encoder/dense/kernel"] + p["mfcc_encoder/dense/bias"])
  if masks.get("enc") is not None: enc = enc * masks["enc"]
  c1 = _lrelu(enc @ p["rnn_module/dense/kernel"] + p["rnn_module/dense/bias"])
  g = "rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/"
  rnn = gru_seq(c1, seq_len, p[g + "gates/kernel"], p[g + "gates/bias"], p[g + "candidate/kernel"], p[g + "candidate/bias"], masks.get("rnn"))
  d = _lrelu(rnn @ p["bfm_coeff_decoder/dense/kernel"] + p["bfm_coeff_decoder/dense/bias"])
  if masks.get("d0") is not None: d = d * masks["d0"]
  d = _lrelu(d @ p["bfm_coeff_decoder/dense_1/kernel"] + p["bfm_coeff_decoder/dense_1/bias"])
  if masks.get("d1") is not None: d = d * masks["d1"]
  out = d @ p["bfm_coeff_decoder/dense_2/kernel"] + p["bfm_coeff_decoder/dense_2/bias"]
  e = ears * torch.tensor([-2.0, -2.0, -2.0, -4.0], dtype=ears.dtype)
  out = out + F.pad(e, (16, 44))
  return out


def vertex_loss(out, bfm_coeffs, seq_len, id_base, ex_base, meanshape, vmask):
  """add_cost_function (bfmnet.py:215-262) without the regulariser.  out [B,T,64]; bfm_coeffs [B,T,>=144]; bases [3n,80] / [3n,64];
  vmask [3n] (10 on mouth vertices)."""
  B, T, _ = out.shape

  def shape(c):      # Shape_formation (:200-213); the re-centring constant cancels in every difference below, kept for fidelity
    fs = c[..., :80] @ id_base.T + c[..., 80:144] @ ex_base.T + meanshape
    return fs - meanshape.reshape(-1, 3).mean(0).repeat(meanshape.numel() // 3)
  pred = shape(torch.cat([bfm_coeffs[..., :80], out], -1))
  true = shape(bfm_coeffs)
  sl = torch.as_tensor(np.asarray(seq_len))
  fm = (torch.arange(T)[None, :] < sl[:, None]).to(out.dtype)
  frame = ((true - pred).abs() * vmask).sum(-1)
  loss = (frame * fm).sum(-1).mean()
  vm = (torch.arange(T - 1)[None, :] < (sl - 1)[:, None]).to(out.dtype)
  vd = (pred[:, 1:] - pred[:, :-1]) - (true[:, 1:] - true[:, :-1])
  video = (vd.abs() * vmask).sum(-1)
  return loss + (video * vm).sum(-1).mean()


def regularised(name):
  """variables carrying kernel_regularizer / weights_regularizer (tinynet.py:10-100): conv and depthwise kernels of MfccNet"""
  return "MfccNet" in name and (name.endswith("/kernel") or name.endswith("depthwise_weights"))


def trainable(name):
  return not (name.endswith("moving_mean") or name.endswith("moving_variance"))


def train_step(params, adam, ears, mfccs, bfm_coeffs, seq_len, masks, model, lr=1e-4, max_grad_norm=50.0, step_t=1):
  """One build_train_op step in float64.  params: {tf_name: ndarray}; adam: {name: (m, v)} or None (zeros); model: dict with
  idBase, exBase, meanshape, vmask (numpy).  Returns dict(loss, loss_data, grads (clipped), global_norm, params (updated, incl.
  moving statistics), adam)."""
  t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
  p = {k: t64(v).requires_grad_(trainable(k)) for k, v in params.items()}
  stats = Stats()
  m = {k: (None if v is None else t64(v)) for k, v in masks.items()}
  out = forward_train(p, t64(ears), t64(mfccs), seq_len, m, stats)
  data = vertex_loss(out, t64(bfm_coeffs), seq_len, t64(model["idBase"]), t64(model["exBase"]), t64(model["meanshape"]).reshape(-1), t64(model["vmask"]).reshape(-1))
  reg = sum(L2_SCALE * 0.5 * (v * v).sum() for k, v in p.items() if regularised(k))
  loss = data + reg
  names = [k for k in p if trainable(k)]
  grads = torch.autograd.grad(loss, [p[k] for k in names])
  gn = torch.sqrt(sum((g * g).sum() for g in grads))
  scale = max_grad_norm / max(float(gn), max_grad_norm)
  new = {k: np.asarray(v, dtype=np.float64).copy() for k, v in params.items()}
  new_adam, clipped = {}, {}
  lr_t = lr * np.sqrt(1 - 0.999 ** step_t) / (1 - 0.9 ** step_t)
  for k, g in zip(names, grads):
    g = g.numpy() * scale
    clipped[k] = g
    m0, v0 = (adam[k] if adam else (np.zeros_like(g), np.zeros_like(g)))
    m1 = 0.9 * m0 + 0.1 * g
    v1 = 0.999 * v0 + 0.001 * g * g
    new[k] = new[k] - lr_t * m1 / (np.sqrt(v1) + 1e-8)
    new_adam[k] = (m1, v1)
  for scope, (mean, var, n) in stats.d.items():
    mm, mv = scope + "/BatchNorm/moving_mean", scope + "/BatchNorm/moving_variance"
    new[mm] = BN_DECAY * new[mm] + (1 - BN_DECAY) * mean.numpy()
    new[mv] = BN_DECAY * new[mv] + (1 - BN_DECAY) * var.numpy() * (n / max(n - 1, 1))     # fused kernel: unbiased estimate
  return {"loss": float(loss), "loss_data": float(data), "out": out.detach().numpy(), "grads": clipped, "global_norm": float(gn),
          "params": new, "adam": new_adam, "stats": {k: (a.numpy(), b.numpy(), n) for k, (a, b, n) in stats.d.items()}}


def synthetic_model(nver=300, seed=0):
  """small stand-in for the external BFM bases (BFM_model_front.mat is not in the repo): idBase [3n,80], exBase [3n,64], meanshape [3n],
  mouth mask 10 on a tenth of the vertices (bfmnet.py:131-134)."""
  rng = np.random.default_rng(seed)
  vm = np.ones((nver, 3))
  vm[rng.choice(nver, nver // 10, replace=False)] = 10.0
  return {"idBase": rng.normal(0, 0.3, (3 * nver, 80)), "exBase": rng.normal(0, 0.3, (3 * nver, 64)),
          "meanshape": rng.normal(0, 1.0, (3 * nver,)), "vmask": vm.reshape(-1)}
