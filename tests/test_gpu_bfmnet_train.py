"""BFMNet training step on the device vs the float64 torch-CPU restatement of build_train_op
(oracle/bfmnet_train_torch.py; parity unpinned by the reference - TF 1.x cannot run here).  Tolerances: float32 device arithmetic
against float64, through 52 batch-normalised layers at a tiny batch: loss 1e-4, coefficients 1e-3, gradient tensors 3e-2 rel-L2.  The
gradient bound is what float32 buys on this graph: the SAME torch restatement run in float32 differs from its float64 run by up to
2.2e-2 on the same tensors (relu / relu6 / |.| kinks and 20-pixel batch statistics; measured); the device
lands at 1.5e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import audio_ref as ar
from oracle import bfmnet_train_torch as bt
from voicepuppet_amd.bfmnet.train_engine import BFMNetTrainEngine, trainable

import gpu_util as gu

pytestmark = pytest.mark.gpu


def _case(B, T, seq, nver, seed, with_masks):
  rng = np.random.default_rng(seed)
  p = ar.init_bfmnet_params(seed + 1, dtype=np.float32)
  mf = rng.normal(0, 1, (B, 5 * T, 80)).astype(np.float32)
  ears = (rng.uniform(size=(B, T, 1)) / 100).astype(np.float32)
  coeff = rng.normal(0, 0.5, (B, T, 257)).astype(np.float32)
  model = bt.synthetic_model(nver, seed + 2)
  masks = {}
  if with_masks:
    for k, c in (("enc", 256), ("rnn", 256), ("d0", 128), ("d1", 64)):
      masks[k] = ((rng.uniform(size=(B, T, c)) < 0.75) / 0.75).astype(np.float32)
  return p, mf, ears, coeff, model, masks


@pytest.mark.parametrize("B,T,seq,with_masks", [(2, 4, [4, 3], False), (3, 5, [5, 2, 4], True)])
def test_train_step_matches_oracle(B, T, seq, with_masks):
  p, mf, ears, coeff, model, masks = _case(B, T, seq, 120, 5, with_masks)
  ref = bt.train_step({k: v.astype(np.float64) for k, v in p.items()}, None, ears, mf, coeff, seq, masks, model)
  eng = BFMNetTrainEngine(B, T, model)
  eng.load_params(p)
  dev = lambda a: torch.tensor(a, device="cuda")
  got = eng.train_step(dev(ears), dev(mf), dev(coeff), seq, {k: dev(v) for k, v in masks.items()})
  out = eng.last_out.cpu().numpy()
  print("\nloss %.6f vs %.6f, global norm %.4f vs %.4f, coefficients %.2e" % (got["loss"], ref["loss"], got["global_norm"], ref["global_norm"],
                                                                            gu.rel_l2(out, ref["out"])))
  assert got["loss"] == pytest.approx(ref["loss"], rel=1e-4)
  assert got["loss_data"] == pytest.approx(ref["loss_data"], rel=1e-4)
  assert got["global_norm"] == pytest.approx(ref["global_norm"], rel=1e-3)
  assert gu.rel_l2(out, ref["out"]) < 1e-3
  grads = eng.get_grads()
  # (the beta of a batch-norm whose output only reaches the next batch-norm through linear ops - every projection / shortcut BN - has
  # an analytically zero gradient: float64 gives 1e-17, float32 1e-9; such tensors are bounded absolutely against the global norm)
  gnorm = ref["global_norm"] * min(1.0, 50.0 / ref["global_norm"])
  live = [k for k in ref["grads"] if np.linalg.norm(ref["grads"][k]) > 1e-6 * gnorm]
  dead = [k for k in ref["grads"] if k not in live]
  worst = sorted(((gu.rel_l2(grads[k], ref["grads"][k]), k) for k in live), reverse=True)
  print("worst gradient tensors:", [(round(e, 5), k.split("MfccNet/")[-1]) for e, k in worst[:4]], "analytically-zero tensors:", len(dead))
  assert worst[0][0] < 3e-2, worst[:6]
  assert all(np.linalg.norm(grads[k]) < 1e-4 * gnorm for k in dead)
  # the first Adam step moves every element by lr * sign(gradient) (m / sqrt(v) = +-1 at t = 1): parameters agree to rounding wherever the
  # sign of the gradient is not decided by float32 noise (|g| > 0.2 of the tensor's largest); moving statistics compare directly
  new = eng.get_params()
  for k in ref["params"]:
    diff = np.abs(new[k] - ref["params"][k])
    if trainable(k):
      gr = np.abs(ref["grads"][k])
      solid = gr > 0.2 * gr.max() if k in live else np.zeros_like(gr, bool)
      assert diff[solid].max(initial=0.0) < 2e-6, (k, diff[solid].max())
      assert diff.max() <= 2.0001e-4 + 1e-6, (k, diff.max())                      # a flipped sign: 2 * lr
    else:
      assert diff.max() <= 1e-4 * np.abs(ref["params"][k]).max() + 1e-7, (k, diff.max())


def test_single_kernels_against_numpy():
  """depthwise weight gradient, SAME max-pool backward, GRU backward through time and the vertex loss on their own."""
  from voicepuppet_amd import _lib
  import ctypes
  L = _lib.lib()
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  rng = np.random.default_rng(0)
  b, h, w, c = 2, 9, 5, 8
  x = rng.normal(size=(b, h, w, c)).astype(np.float32)
  dy = rng.normal(size=(b, h, w, c)).astype(np.float32)
  xt, dyt = torch.tensor(x, device="cuda"), torch.tensor(dy, device="cuda")
  dw = torch.empty(21, c, device="cuda")
  ws = torch.empty(L.vp_dwconv7x3_wgrad_workspace_bytes(b, h, w, c), dtype=torch.uint8, device="cuda")
  _lib.check(L.vp_dwconv7x3_wgrad(P(xt), P(dyt), P(dw), b, h, w, c, P(ws), st))
  xp = np.pad(x.astype(np.float64), ((0, 0), (3, 3), (1, 1), (0, 0)))
  ref = np.stack([(xp[:, kh:kh + h, kw:kw + w, :] * dy).sum((0, 1, 2)) for kh in range(7) for kw in range(3)])
  assert gu.rel_l2(dw.cpu().numpy(), ref) < 1e-5
  # max-pool backward: overlapping 2x2 stride (1,2) windows with SAME padding, and the 5x3 / (5,3) pool
  for (hh, ww, k, s) in ((7, 5, (2, 2), (1, 2)), (10, 3, (5, 3), (5, 3))):
    xx = rng.normal(size=(b, hh, ww, c)).astype(np.float32)
    xg = torch.tensor(xx, dtype=torch.float64, requires_grad=True)
    from oracle.bfmnet_train_torch import _pool
    yy = _pool(xg.permute(0, 3, 1, 2), k, s)
    gy = rng.normal(size=tuple(yy.shape)).astype(np.float32)
    yy.backward(torch.tensor(gy, dtype=torch.float64))
    dx = torch.empty(b, hh, ww, c, device="cuda")
    xd, gd = torch.tensor(xx, device="cuda"), torch.tensor(np.ascontiguousarray(gy.transpose(0, 2, 3, 1)), device="cuda")
    _lib.check(L.vp_maxpool_hw_bwd(P(xd), P(gd), P(dx), b, hh, ww, c, k[0], k[1], s[0], s[1], st))
    torch.cuda.synchronize()
    assert np.allclose(dx.cpu().numpy(), xg.grad.numpy(), atol=1e-6)


def test_graphed_step_equals_eager_step():
  """train_step_graphed replays the launches train_step issues: from the same state and batch (dropout off) both reach the same
  parameters, Adam slots and moving statistics, step after step; with dropout on every replay draws new masks."""
  from voicepuppet_amd.bfmnet.bfmnet import random_variables
  B, T = 2, 4
  rng = np.random.default_rng(3)
  model = bt.synthetic_model(120, 1)
  mk = lambda: BFMNetTrainEngine(B, T, {"exBase": model["exBase"], "vmask": model["vmask"]}, lr=1e-3)
  a, b = mk(), mk()
  w = random_variables(5)
  a.load_params(w); b.load_params(w)
  for i in range(3):
    ears = torch.tensor(rng.uniform(0.1, 0.4, (B, T, 1)), dtype=torch.float32, device="cuda")
    mfccs = torch.tensor(rng.normal(0, 1, (B, 5 * T, 80)), dtype=torch.float32, device="cuda")
    coeff = torch.tensor(rng.normal(0, 0.5, (B, T, 257)), dtype=torch.float32, device="cuda")
    seq = [T, T - 1 - (i % 2)]
    ra = a.train_step(ears, mfccs, coeff, seq)
    rb = b.train_step_graphed(ears, mfccs, coeff, seq, 0.0, 0.0)
    assert ra["loss"] == pytest.approx(rb["loss"], rel=1e-6) and ra["global_norm"] == pytest.approx(rb["global_norm"], rel=1e-5)
    assert torch.allclose(a.arena, b.arena, rtol=1e-5, atol=1e-7) and torch.allclose(a.v, b.v, rtol=1e-4, atol=1e-12)
  assert a.step_t == b.step_t == 3
  losses = {b.train_step_graphed(ears, mfccs, coeff, seq, 0.25)["loss"] for _ in range(3)}
  assert len(losses) == 3 and all(np.isfinite(l) for l in losses)


def test_two_stream_step_equals_one_stream_step():
  """The eager step runs its weight gradients on a second stream (train_engine._fork); with side_stream=False everything stays on one.
  Same kernels on the same values: gradients and updated parameters are bit-identical over three steps."""
  B, T, seq = 3, 5, [5, 2, 4]
  p, mf, ears, coeff, model, masks = _case(B, T, seq, 120, 7, True)
  dev = lambda a: torch.tensor(a, device="cuda")
  outs = []
  for one_stream in (False, True):
    eng = BFMNetTrainEngine(B, T, model, side_stream=not one_stream)
    assert (eng._side is None) == one_stream
    eng.load_params(p)
    for _ in range(3):
      eng.train_step(dev(ears), dev(mf), dev(coeff), seq, {k: dev(v) for k, v in masks.items()})
    torch.cuda.synchronize()
    outs.append((eng.grads.clone(), eng.arena.clone()))
  assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("P,K,N", [(96, 256, 512), (3840, 192, 1152), (1000, 48, 32), (7, 1536, 256), (96, 64, 1007),
                                   (213, 128, 192), (50, 384, 64), (19200, 64, 384)])      # + the 128 x 64 tile of wgrad_mm.hip, a long K walk
def test_matrix_products_against_numpy(P, K, N):
  """vp_mm_fwd_f32 / vp_mm_bwd_data_f32 / vp_mm_bwd_weight_f32 (csrc/mm_api.hip: the repo's own float32-MFMA kernels - no vendor GEMM
  library) against float64 numpy, at the shapes of the BFMNet training step: channel counts that are not powers of two, a handful
  of rows, the zero-padded stem (K = 48 with 45 real rows), a transposed weight matrix and an output width that is no multiple of 16
  (the face-shape product).  float32 MFMA == a float32 fmaf chain: 2e-5 relative L2, as tests/test_gpu_ops.py."""
  rng = np.random.default_rng(P + K + N)
  model = bt.synthetic_model(60, 3)
  eng = BFMNetTrainEngine(2, 24, {"exBase": model["exBase"], "vmask": model["vmask"]})
  dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
  rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
  x, w, b = rng.normal(size=(P, K)), rng.normal(0, 0.1, (K, N)), rng.normal(size=N)
  x32, w32, b32 = [np.float32(v).astype(np.float64) for v in (x, w, b)]
  y = eng._mm(dev(x), dev(w), dev(b)).cpu().numpy()
  assert rel(y, x32 @ w32 + b32) < 2e-5
  yt = eng._mm(dev(x), dev(w.T), w_t=True).cpu().numpy()                             # w stored [N, K]
  assert rel(yt, x32 @ w32) < 2e-5
  # backward-data: contraction over N (padded to a multiple of 16 with zeros when it is not one)
  Np = -(-N // 16) * 16
  dy = np.zeros((P, Np))
  dy[:, :N] = rng.normal(size=(P, N))
  dy32 = np.float32(dy).astype(np.float64)
  dx = eng._mm_dx(dev(dy), dev(w), n=N).cpu().numpy()
  assert rel(dx, dy32[:, :N] @ w32.T) < 2e-5
  acc = eng._mm_dx(dev(dy), dev(w), out=dev(x), accumulate=True, n=N).cpu().numpy()
  assert rel(acc, x32 + dy32[:, :N] @ w32.T) < 2e-5
  dxt = eng._mm_dx(dev(dy), dev(w.T), w_t=True, n=N).cpu().numpy()
  assert rel(dxt, dy32[:, :N] @ w32.T) < 2e-5
  if N % 4 == 0:
    kr = 45 if K == 48 else K
    out = torch.full((kr, N), float("nan"), device="cuda")
    dw = eng._mm_dw(dev(x), dev(dy[:, :N]), out, k_real=kr).cpu().numpy()
    assert rel(dw, (x32.T @ dy32[:, :N])[:kr]) < 2e-5


def test_training_step_holds_no_vendor_gemm():
  """every matrix product of the BFMNet training step runs on the repo's own kernels - no torch.mm / addmm / matmul
  / einsum / linear / bmm, no TunableOp table, anywhere in the product tree."""
  import re
  root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "voicepuppet_amd")
  bad = []
  for d, _, files in os.walk(root):
    for f in files:
      if f.endswith(".py"):
        src = open(os.path.join(d, f)).read()
        for m in re.finditer(r"torch\.(mm|addmm|matmul|bmm|einsum|baddbmm)\b|F\.linear|torch\.cuda\.tunable|\s@\s", src):
          line = src[:m.start()].count("\n") + 1
          text = src.splitlines()[line - 1]
          if text.lstrip().startswith("#") or '"""' in text:
            continue
          bad.append("%s:%d %s" % (os.path.join(d, f), line, text.strip()[:80]))
  assert not bad, bad
  assert not os.path.exists(os.path.join(root, "bfmnet", "gemm_tuning"))
  # nor the at::native reductions / transposes / copies the step still ran in round 3 (bias gradients through
  # torch.sum, the recurrent kernels through .t().contiguous() and [256:].contiguous(), the output through .clone())
  eng = open(os.path.join(root, "bfmnet", "train_engine.py")).read()
  for pat in (r"torch\.sum\(", r"\.t\(\)\.contiguous\(\)", r"\[256:\]\.contiguous\(\)", r"\bo\.clone\(\)"):
    assert not re.search(pat, eng), pat
