"""CPU checks of the BFMNet-training oracle (oracle/bfmnet_train_torch.py).  Parity with TensorFlow itself is unpinned
(no TF here, no vectors in the reference); what can be pinned on the CPU is internal consistency:
  * the training-mode forward equals the numpy inference restatement (oracle/audio_ref.py, itself pinned by the round-1 golden vectors)
    when the moving statistics are set to the batch statistics of the same batch;
  * vertex_loss equals a literal, loop-level restatement of add_cost_function (bfmnet.py:229-271) that builds both face shapes in full;
  * autograd's gradient agrees with central differences of the whole loss on sampled coordinates;
  * clip_by_global_norm / Adam / moving-average arithmetic obey their defining identities."""
import numpy as np
import pytest
import torch

from oracle import audio_ref as ar
from oracle import bfmnet_train_torch as bt


def _case(B=2, T=2, nver=40, seed=0, seq=(2, 1), big=1.0):
  rng = np.random.default_rng(seed)
  p = ar.init_bfmnet_params(seed=seed)
  ears = rng.uniform(0.1, 0.4, (B, T, 1))
  mfccs = rng.normal(0, 1, (B, 5 * T, 80))
  coeff = rng.normal(0, 0.5 * big, (B, T, 257))
  model = bt.synthetic_model(nver, seed)
  return p, ears, mfccs, coeff, list(seq), model


def test_training_forward_equals_inference_with_batch_statistics():
  p, ears, mfccs, coeff, seq, model = _case()
  r = bt.train_step(p, None, ears, mfccs, coeff, seq, {}, model)
  q = dict(p)
  for scope, (mean, var, n) in r["stats"].items():
    q[scope + "/BatchNorm/moving_mean"] = mean
    q[scope + "/BatchNorm/moving_variance"] = var
  out = ar.bfmnet_fwd(q, ears, mfccs, np.asarray(seq))["BFMCoeffDecoder"]
  assert np.abs(out - r["out"]).max() < 1e-10


def test_vertex_loss_against_literal_restatement():
  p, ears, mfccs, coeff, seq, model = _case(B=2, T=3, seq=(3, 2))
  rng = np.random.default_rng(5)
  out = rng.normal(0, 0.5, (2, 3, 64))
  got = float(bt.vertex_loss(torch.tensor(out), torch.tensor(coeff), seq, torch.tensor(model["idBase"]), torch.tensor(model["exBase"]),
                             torch.tensor(model["meanshape"]), torch.tensor(model["vmask"])))
  idb, exb, ms, vm = model["idBase"], model["exBase"], model["meanshape"], model["vmask"]
  centre = ms.reshape(-1, 3).mean(0)

  def shape(c):   # Shape_formation (bfmnet.py:215-227)
    s = idb @ c[:80] + exb @ c[80:144] + ms
    return (s.reshape(-1, 3) - centre).reshape(-1)
  B, T = 2, 3
  tmax = max(seq)
  true = np.array([[shape(coeff[b, t]) for t in range(T)] for b in range(B)])
  pred = np.array([[shape(np.concatenate([coeff[b, t, :80], out[b, t]])) for t in range(T)] for b in range(B)])
  frame = sum(np.sum(np.abs(true[b, t] - pred[b, t]) * vm) for b in range(B) for t in range(tmax) if t < seq[b]) / B
  video = 0.0
  for b in range(B):
    for t in range(tmax - 1):
      if t < seq[b] - 1:
        video += np.sum(np.abs((pred[b, t + 1] - pred[b, t]) - (true[b, t + 1] - true[b, t])) * vm)
  want = frame + video / B
  assert abs(got - want) < 1e-9 * max(1.0, abs(want))


def test_autograd_against_central_differences():
  p, ears, mfccs, coeff, seq, model = _case(seed=3)
  r = bt.train_step(p, None, ears, mfccs, coeff, seq, {}, model, max_grad_norm=1e30)   # no clipping: grads are the raw gradient

  def loss_of(q):
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    pp = {k: t64(v) for k, v in q.items()}
    out = bt.forward_train(pp, t64(ears), t64(mfccs), seq, {}, bt.Stats())
    data = bt.vertex_loss(out, t64(coeff), seq, t64(model["idBase"]), t64(model["exBase"]), t64(model["meanshape"]).reshape(-1),
                          t64(model["vmask"]).reshape(-1))
    return float(data + sum(bt.L2_SCALE * 0.5 * (v * v).sum() for k, v in pp.items() if bt.regularised(k)))
  rng = np.random.default_rng(0)
  names = ["mfcc_encoder/MfccNet/block0_0/conv2d/conv2d/kernel", "mfcc_encoder/MfccNet/block3_0/depthwise_conv2d/SeparableConv2d/depthwise_weights",
           "mfcc_encoder/MfccNet/block5_0/expansion_1x1_conv2d/BatchNorm/beta", "rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel",
           "bfm_coeff_decoder/dense_1/kernel", "mfcc_encoder/dense/bias"]
  for n in names:
    g = r["grads"][n]
    idx = np.unravel_index(int(np.argmax(np.abs(g))), g.shape) if rng.random() < 0.5 else tuple(int(rng.integers(s)) for s in g.shape)
    h = 1e-7
    q = {k: np.array(v, dtype=np.float64) for k, v in p.items()}
    q[n][idx] += h
    up = loss_of(q)
    q[n][idx] -= 2 * h
    dn = loss_of(q)
    fd = (up - dn) / (2 * h)
    # the loss is piecewise smooth (abs, relu, relu6, max-pool): an early kernel reaches thousands of units, so a kink inside +-h is
    # possible and bounds the agreement; a wrong backward formula would be off by O(1)
    assert abs(fd - g[idx]) < 5e-3 * max(1.0, abs(g[idx])), (n, idx, fd, g[idx])


def test_clip_adam_and_moving_average_identities():
  p, ears, mfccs, coeff, seq, model = _case(seed=4, big=4.0)
  raw = bt.train_step(p, None, ears, mfccs, coeff, seq, {}, model, max_grad_norm=1e30)
  clip = raw["global_norm"] / 3.0
  r = bt.train_step(p, None, ears, mfccs, coeff, seq, {}, model, max_grad_norm=clip, lr=1e-3)
  assert abs(r["global_norm"] - raw["global_norm"]) < 1e-9 * raw["global_norm"]
  norm = np.sqrt(sum(float((g * g).sum()) for g in r["grads"].values()))
  assert abs(norm - clip) < 1e-9 * clip
  for k, g in r["grads"].items():
    assert np.allclose(g, raw["grads"][k] / 3.0, rtol=1e-9, atol=1e-14)
    # first Adam step: m/(sqrt(v)+eps) = g/(|g| + eps/sqrt(1-beta2)...) -> a step of lr * sign(g) wherever |g| >> 1e-8
    big = np.abs(g) > 1e-2
    step = p[k] - r["params"][k]
    assert np.allclose(step[big], 1e-3 * np.sign(g[big]), rtol=1e-3)
  # a second step from the returned state equals the closed form of tf.train.AdamOptimizer (lr_t with both beta powers)
  r2 = bt.train_step(r["params"], r["adam"], ears, mfccs, coeff, seq, {}, model, max_grad_norm=clip, lr=1e-3, step_t=2)
  k = "bfm_coeff_decoder/dense_2/kernel"
  m1, v1 = r["adam"][k]
  g2 = r2["grads"][k]
  m2, v2 = 0.9 * m1 + 0.1 * g2, 0.999 * v1 + 0.001 * g2 * g2
  lr_t = 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
  assert np.allclose(r2["params"][k], r["params"][k] - lr_t * m2 / (np.sqrt(v2) + 1e-8), rtol=1e-12, atol=1e-15)
  # moving statistics: decay 0.999 towards the batch mean / the UNBIASED batch variance; untouched by the optimiser
  scope = "mfcc_encoder/MfccNet/block0_0/conv2d"
  mean, var, n = r["stats"][scope]
  assert np.allclose(r["params"][scope + "/BatchNorm/moving_mean"], 0.999 * p[scope + "/BatchNorm/moving_mean"] + 0.001 * mean)
  assert np.allclose(r["params"][scope + "/BatchNorm/moving_variance"], 0.999 * p[scope + "/BatchNorm/moving_variance"] + 0.001 * var * n / (n - 1))
  assert scope + "/BatchNorm/moving_mean" not in r["grads"]


def test_sequence_mask_blocks_gradient_of_padded_frames():
  """Frames at or beyond seq_len contribute nothing: changing their targets changes neither the loss nor any gradient."""
  p, ears, mfccs, coeff, seq, model = _case(seed=6, seq=(2, 1))
  a = bt.train_step(p, None, ears, mfccs, coeff, seq, {}, model)
  coeff2 = coeff.copy()
  coeff2[1, 1, 80:144] += 3.0
  b = bt.train_step(p, None, ears, mfccs, coeff2, seq, {}, model)
  assert a["loss"] == pytest.approx(b["loss"], rel=1e-13)
  for k in a["grads"]:
    assert np.allclose(a["grads"][k], b["grads"][k], rtol=1e-12, atol=1e-15)
