"""BFMNet.build_train_op on the device (reference: voicepuppet/bfmnet/bfmnet.py:215-323 over tinynet.py:7-212).

One `train_step` = training-mode forward (batch-statistics batch_norm, dropout masks), the vertex-space loss of add_cost_function
plus the l2 regulariser, the full backward pass, clip_by_global_norm and one tf.train.AdamOptimizer update, plus the moving-average
update of every batch_norm.  Division of labour: everything that is a plain matrix product (1x1 convolutions and dense layers on
[pixels, channels] matrices, the GRU's input / recurrent weight gradients, the [B*T, 64] x [64, 3n] face-shape products) is a
rocBLAS GEMM through torch.mm; everything else - batch-norm statistics and backward, activations and dropout, the depthwise 7x3
convolution with its two gradients, SAME max-pools, the im2col of the stem, the GRU recurrence forward and backward through time, the
loss with its gradient, the sums of squares, Adam - is a hand-written HIP kernel of libvp_hip.so (csrc/bfm_train.hip,
audio_kernels.hip, pointwise.hip).  No CPU fallback: the constructor raises without a GPU.

Layouts: activations NHWC float32 viewed as [pixels, channels]; parameters live in ONE flat float32 arena (trainable variables first,
moving statistics behind them) in the order of oracle/audio_ref.bfmnet_manifest() = the TF variable order, so the Adam slots and the
gradient arena are flat too, the batch statistics of every batch-norm land in one flat buffer laid out like the arena's moving-statistics
tail (one kernel updates all 114 moving averages), and regulariser / clipping / Adam are three launches over the whole arena.

The step's shapes are static, so `train_step_graphed` captures the ~900 launches of one step (dropout draws included) into a hipGraph
once and replays it: the per-step host work is four small copies into the graph's input buffers and one read of three scalars."""
import ctypes
import math
import os

import numpy as np
import torch

from .. import _lib
from ..audio import bfmnet_manifest

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_RELU6 = 0, 1, 2, 5
BN_EPS, BN_DECAY, L2_SCALE = 1e-3, 0.999, 1e-4
_P = ctypes.c_void_p

# (scope, out_channels, expansion, pool_after)   tinynet.py:172-203
BLOCKS = [("block1_0", 64, 1, False), ("block2_0", 64, 6, True), ("block2_1", 64, 6, False), ("block3_0", 128, 6, True),
          ("block3_1", 128, 6, False), ("block3_2", 128, 6, False), ("block4_0", 192, 6, True), ("block4_1", 192, 6, False),
          ("block4_2", 192, 6, False), ("block4_3", 192, 6, False), ("block5_0", 256, 6, False), ("block5_1", 256, 6, False),
          ("block5_2", 256, 6, False), ("block6_0", 256, 6, True), ("block6_1", 256, 6, False), ("block6_2", 256, 6, False),
          ("block7_0", 256, 6, False)]
PREFIX = "mfcc_encoder/MfccNet/"
GRU = "rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/"


def _ptr(t):
  return _P(t.data_ptr()) if t is not None else _P(0)


def _stream():
  return _P(torch.cuda.current_stream().cuda_stream)


def trainable(name):
  return not (name.endswith("moving_mean") or name.endswith("moving_variance"))


def regularised(name):
  return "MfccNet" in name and (name.endswith("/kernel") or name.endswith("depthwise_weights"))


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning", "gfx950_batch4_batch32.csv")


def use_tuned_gemms(path=TUNED_GEMMS):
  """The step's ~170 f32 GEMMs are skinny (K = 32 ... 256 against 10^4 ... 10^5 rows) and rocBLAS's default heuristic picks solutions
  that run at a third of the f32 MFMA peak; PyTorch's TunableOp looks each shape up in a results file (rocBLAS / hipBLASLt solution per
  shape, searched once on an MI355X by scripts/tune_bfmnet_gemms.sh for the 24-frame clips at batch 4 and 32: 9.4 -> 7.9 ms and
  31.8 -> 23.1 ms per step).  Shapes that are not in the file, or a file written for another ROCm build (its validator lines do not
  match), fall back to the default solutions.  Process-wide: TunableOp is a torch global."""
  if os.environ.get("VP_NO_TUNED_GEMMS") or not os.path.exists(path):
    return False
  tun = torch.cuda.tunable
  tun.enable(True)
  tun.tuning_enable(False)
  return bool(tun.read_file(path))


class BFMNetTrainEngine:
  def __init__(self, batch, frames, model, lr=1e-4, max_grad_norm=50.0, num_mel_bins=80, tuned_gemms=True):
    """model: dict with exBase [3n,64] and vmask [3n] (the mouth-weighted vertex mask of bfmnet.py:131-134); idBase / meanshape cancel in
    every term of the loss (both face shapes share the identity coefficients) and are not needed on the device."""
    if not torch.cuda.is_available():
      raise RuntimeError("BFMNetTrainEngine needs an MI355X (no CPU fallback)")
    self.L = _lib.lib()
    self.tuned_gemms = use_tuned_gemms() if tuned_gemms else False
    self.B, self.T, self.W0 = batch, frames, num_mel_bins
    self.lr, self.clip = lr, max_grad_norm
    dev = torch.device("cuda", torch.cuda.current_device())
    self.dev = dev
    man = [(n, s) for n, _, s in bfmnet_manifest()]
    order = [(n, s) for n, s in man if trainable(n)] + [(n, s) for n, s in man if not trainable(n)]
    self.ntrain = sum(int(np.prod(s)) for n, s in order if trainable(n))
    assert self.ntrain % 4 == 0
    total = sum(int(np.prod(s)) for n, s in order)
    self.arena = torch.zeros(total, dtype=torch.float32, device=dev)
    self.grads = torch.zeros(self.ntrain, dtype=torch.float32, device=dev)
    self.m = torch.zeros(self.ntrain, dtype=torch.float32, device=dev)
    self.v = torch.zeros(self.ntrain, dtype=torch.float32, device=dev)
    self.bstats = torch.zeros(total - self.ntrain, dtype=torch.float32, device=dev)     # batch mean / variance, laid out like the arena's tail
    self.bfactor = None                                                                 # (1 - decay) [* n / (n - 1) on variance slots]
    self._bn_rows = {}
    l2 = np.zeros(self.ntrain, np.float32)
    self.p, self.g, self.bs, self.shapes = {}, {}, {}, {}
    off = 0
    for n, s in order:
      k = int(np.prod(s))
      self.p[n] = self.arena[off:off + k].view(s)
      if trainable(n):
        self.g[n] = self.grads[off:off + k].view(s)
        if regularised(n):
          l2[off:off + k] = 1.0
      else:
        self.bs[n] = self.bstats[off - self.ntrain:off - self.ntrain + k]
      self.shapes[n] = s
      off += k
    self.l2mask = torch.from_numpy(l2).to(dev)
    self.step_t = 0
    self.lr_t = torch.zeros(1, dtype=torch.float32, device=dev)
    self.exbase = torch.tensor(np.asarray(model["exBase"], dtype=np.float32), device=dev).contiguous()       # [3n, 64]
    self.vmask = torch.tensor(np.asarray(model["vmask"], dtype=np.float32).reshape(-1), device=dev).contiguous()
    self.J = self.exbase.shape[0]
    self.ears_scale = torch.tensor([-2.0, -2.0, -2.0, -4.0], device=dev)
    self._ws = {}
    self._graphs = {}

  def tune_gemms(self, ears, mfccs, bfm_coeffs, seq_len, path, max_ms_per_shape=20):
    """Search the GEMM solutions for THIS batch / clip length (one gradient-only step with TunableOp's tuning on, a few minutes) and
    write them to `path` in TunableOp's format; later engines pick them up with use_tuned_gemms(path)."""
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.set_max_tuning_duration(max_ms_per_shape)
    tun.tuning_enable(True)
    try:
      self.train_step(ears, mfccs, bfm_coeffs, seq_len, apply=False)
      torch.cuda.synchronize()
    finally:
      tun.tuning_enable(False)
    with open(path, "w") as f:                                                     # TunableOp's own CSV layout
      for k, v in tun.get_validators():
        f.write("Validator,%s,%s\n" % (k, v))
      for row in tun.get_results():
        f.write(",".join(str(x) for x in row) + "\n")
    return path

  # ---- parameters -----------------------------------------------------------------------------------------------------------
  def load_params(self, params):
    for n, t in self.p.items():
      if n in params:
        t.copy_(torch.tensor(np.asarray(params[n], dtype=np.float32).reshape(self.shapes[n]), device=self.dev))

  def trainables(self):
    """[(name, shape)] in the order of the flat gradient / Adam buffers."""
    return [(n, self.shapes[n]) for n in self.p if trainable(n)]

  def get_params(self):
    return {n: t.detach().cpu().numpy().copy() for n, t in self.p.items()}

  def get_grads(self, unclipped=False):
    """Gradients of the last step.  The arena holds them after clip_by_global_norm (the Adam kernel scales in place);
    unclipped=True undoes that scale, i.e. returns what tf.gradients / compute_gradients hand out (bfmnet.py:300-303: the
    reference's nodes['Grads'] are taken before the clip)."""
    scale = 1.0
    if unclipped and getattr(self, "_last_ss", None) is not None:
      gn = float(torch.sqrt(self._last_ss))
      scale = max(gn, self.clip) / self.clip
    return {n: t.detach().cpu().numpy().copy() * np.float32(scale) for n, t in self.g.items()}

  # ---- kernel wrappers ------------------------------------------------------------------------------------------------------
  def _work(self, key, nbytes):
    w = self._ws.get(key)
    if w is None or w.numel() < nbytes:
      if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("workspace %s must exist before a graph capture (run one eager step first)" % key)
      w = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
      self._ws[key] = w
    return w

  def _bn_fwd(self, y, scope):
    """batch statistics of y [P, C] -> (mean, rstd, shift); mean / biased variance go to the flat batch-statistics buffer."""
    P, C = y.shape
    self._bn_rows[scope] = P
    mean, var = self.bs[scope + "/BatchNorm/moving_mean"], self.bs[scope + "/BatchNorm/moving_variance"]
    rs = torch.empty(3, C, dtype=torch.float32, device=self.dev)
    ws = self._work("bn", self.L.vp_bn_train_workspace_bytes(P, C))
    _lib.check(self.L.vp_bn_train_fwd(_ptr(y), P, C, _ptr(self.p[scope + "/BatchNorm/beta"]), BN_EPS, _ptr(mean), _ptr(var), _ptr(rs[0]), _ptr(rs[1]),
                                      _ptr(rs[2]), _ptr(ws), _stream()), "vp_bn_train_fwd")
    return mean, rs[0], rs[2]

  def _bn_act_bwd(self, da, y, mean, rstd, shift, act, scope):
    """backward of act(batch_norm(y)): the activation's derivative is recomputed from y inside the kernels."""
    P, C = y.shape
    dx = torch.empty_like(y)
    ws = self._work("bn", self.L.vp_bn_train_workspace_bytes(P, C))
    _lib.check(self.L.vp_bn_act_train_bwd(_ptr(y), _ptr(da), P, C, _ptr(mean), _ptr(rstd), _ptr(shift), act, _ptr(dx),
                                          _ptr(self.g[scope + "/BatchNorm/beta"]), _ptr(ws), _stream()), "vp_bn_act_train_bwd")
    return dx

  def _act(self, x, scale, shift, act, mask=None):
    P, C = x.shape
    y = torch.empty_like(x)
    _lib.check(self.L.vp_affine_act_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(mask), P, C, act, _ptr(y), _stream()), "vp_affine_act_fwd")
    return y

  def _act_bwd(self, dy, ya, act, mask=None):
    dx = torch.empty_like(dy)
    _lib.check(self.L.vp_act_bwd(_ptr(dy), _ptr(ya), _ptr(mask), dy.numel(), act, _ptr(dx), _stream()), "vp_act_bwd")
    return dx

  def _dw(self, x2d, w21, H, W):
    C = x2d.shape[1]
    y = torch.empty_like(x2d)
    _lib.check(self.L.vp_dwconv7x3_raw(_ptr(x2d), _ptr(w21), _ptr(y), self.B, H, W, C, _stream()), "vp_dwconv7x3_raw")
    return y

  def _dw_wgrad(self, x2d, dy2d, H, W, out):
    C = x2d.shape[1]
    ws = self._work("dw", self.L.vp_dwconv7x3_wgrad_workspace_bytes(self.B, H, W, C))
    _lib.check(self.L.vp_dwconv7x3_wgrad(_ptr(x2d), _ptr(dy2d), _ptr(out), self.B, H, W, C, _ptr(ws), _stream()), "vp_dwconv7x3_wgrad")

  def _pool(self, x2d, H, W, k, s):
    C = x2d.shape[1]
    Ho, Wo = -(-H // s[0]), -(-W // s[1])
    y = torch.empty(self.B * Ho * Wo, C, dtype=torch.float32, device=self.dev)
    _lib.check(self.L.vp_maxpool_hw(_ptr(x2d), _ptr(y), self.B, H, W, C, k[0], k[1], s[0], s[1], _stream()), "vp_maxpool_hw")
    return y, Ho, Wo

  def _pool_bwd(self, x2d, dy2d, H, W, k, s):
    C = x2d.shape[1]
    dx = torch.empty_like(x2d)
    _lib.check(self.L.vp_maxpool_hw_bwd(_ptr(x2d), _ptr(dy2d), _ptr(dx), self.B, H, W, C, k[0], k[1], s[0], s[1], _stream()), "vp_maxpool_hw_bwd")
    return dx

  def _sumsq(self, x):
    n = self.L.vp_sumsq_partials(x.numel())
    part = torch.empty(n, dtype=torch.float64, device=self.dev)
    _lib.check(self.L.vp_sumsq(_ptr(x), x.numel(), _ptr(part), _stream()), "vp_sumsq")
    return part.sum()

  # ---- conv + batch-norm + activation, forward / backward -----------------------------------------------------------------------
  def _cba_fwd(self, x, kernel, bn_scope, act, tape):
    """x [P, cin] @ kernel [cin, cout] -> batch_norm -> act.  tape gets what the backward needs."""
    y = torch.mm(x, kernel)
    mean, rstd, shift = self._bn_fwd(y, bn_scope)
    a = self._act(y, rstd, shift, act)
    tape.append(("cba", x, kernel, y, mean, rstd, shift, act, bn_scope))
    return a

  def _grad2d(self, kernel_view):
    """the gradient slot of the variable a [cin, cout] view of the arena belongs to, as a matrix of the same shape (GEMM output)"""
    name = self._by_offset()[kernel_view.storage_offset()]
    return self.g[name].view(kernel_view.shape)

  def _by_offset(self):
    if not hasattr(self, "_off"):
      self._off = {t.storage_offset(): n for n, t in self.p.items()}
    return self._off

  def _moving_factor(self):
    if self.bfactor is None:
      f = torch.full_like(self.bstats, 1 - BN_DECAY)
      for scope, n in self._bn_rows.items():
        v = self.bs[scope + "/BatchNorm/moving_variance"]
        off = v.storage_offset() - self.bstats.storage_offset()
        f[off:off + v.numel()] = (1 - BN_DECAY) * n / max(n - 1, 1)                                  # fused kernel: unbiased estimate
      self.bfactor = f
    return self.bfactor

  # ---- the step -------------------------------------------------------------------------------------------------------------
  def train_step(self, ears, mfccs, bfm_coeffs, seq_len, masks=None, apply=True):
    """ears [B,T,1], mfccs [B,5T,80], bfm_coeffs [B,T,>=144] (device float32 tensors); seq_len: list / int32 tensor [B];
    masks: optional dict 'enc' [B,T,256], 'rnn' [B,T,256], 'd0' [B,T,128], 'd1' [B,T,64] with entries 0 or 1/keep_prob (dropout draws).
    Returns dict(loss, loss_data, global_norm) of python floats.  apply=False: gradients only (self.g, clipped), no update."""
    seq = torch.as_tensor(seq_len, dtype=torch.int32, device=self.dev).contiguous()
    if apply:
      self._advance()
    res = self._body(ears, mfccs, bfm_coeffs, seq, masks or {}, apply)
    loss, loss_data, gn = res.tolist()
    return {"loss": loss, "loss_data": loss_data, "global_norm": gn}

  def train_step_graphed(self, ears, mfccs, bfm_coeffs, seq_len, drop_rate=0.25, inner_rate=0.25):
    """The same step (apply=True) replayed from a hipGraph captured on first use; the dropout masks are drawn inside the graph
    (draw_masks) from torch's device generator.  One eager step's worth of scratch stays resident in the graph's private pool."""
    key = (float(drop_rate or 0), float(inner_rate or 0))
    g = self._graphs.get(key)
    if g is None:
      g = self._capture(ears, mfccs, bfm_coeffs, seq_len, *key)
      self._graphs[key] = g
    graph, s_in, res = g
    s_in[0].copy_(ears.reshape(s_in[0].shape)); s_in[1].copy_(mfccs); s_in[2].copy_(bfm_coeffs)
    s_in[3].copy_(torch.as_tensor(seq_len, dtype=torch.int32), non_blocking=False)
    self._advance()
    graph.replay()
    loss, loss_data, gn = res.tolist()
    return {"loss": loss, "loss_data": loss_data, "global_norm": gn}

  def _advance(self):
    self.step_t += 1
    self.lr_t.fill_(self.lr * math.sqrt(1 - 0.999 ** self.step_t) / (1 - 0.9 ** self.step_t))

  def _capture(self, ears, mfccs, bfm_coeffs, seq_len, drop_rate, inner_rate):
    s_in = [torch.empty_like(ears.to(self.dev, torch.float32).contiguous()), torch.empty_like(mfccs.to(self.dev, torch.float32).contiguous()),
            torch.empty_like(bfm_coeffs.to(self.dev, torch.float32).contiguous()), torch.zeros(self.B, dtype=torch.int32, device=self.dev)]
    s_in[0].copy_(ears); s_in[1].copy_(mfccs); s_in[2].copy_(bfm_coeffs); s_in[3].copy_(torch.as_tensor(seq_len, dtype=torch.int32))
    # warm-up on a side stream (allocator, rocBLAS handles, workspaces) without touching parameters or optimiser state
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(2):
        self._body(s_in[0], s_in[1], s_in[2], s_in[3], self.draw_masks(drop_rate, inner_rate), False)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    self._moving_factor()                                                            # built from the row counts the warm-up recorded
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      res = self._body(s_in[0], s_in[1], s_in[2], s_in[3], self.draw_masks(drop_rate, inner_rate), True)
    return graph, s_in, res

  def _body(self, ears, mfccs, bfm_coeffs, seq, masks, apply):
    """Every launch of one step, on the current stream, without host synchronisation (capturable).  Returns a float64 device tensor
    [loss, data loss, global norm]."""
    B, T, L, p, g = self.B, self.T, self.L, self.p, self.g
    H, W = 5 * T, self.W0
    masks = masks or {}
    mk = lambda k, c: (masks[k].reshape(B * T, c).contiguous() if masks.get(k) is not None else None)
    tape = []
    # stem: 9x5 stride (1,2) as im2col + GEMM
    Wo = (W + 1) // 2
    col = torch.empty(B * H * Wo, 48, dtype=torch.float32, device=self.dev)
    _lib.check(L.vp_stem_im2col(_ptr(mfccs.contiguous()), _ptr(col), B, H, W, _stream()), "vp_stem_im2col")
    s0 = PREFIX + "block0_0/conv2d"
    k0 = torch.zeros(48, 32, dtype=torch.float32, device=self.dev)
    k0[:45] = p[s0 + "/conv2d/kernel"].reshape(45, 32)
    net = self._cba_fwd(col, k0, s0, ACT_RELU, tape)
    W = Wo
    for scope, cout, exp, pool in BLOCKS:
      b = PREFIX + scope
      cin = net.shape[1]
      inp = net
      a = self._cba_fwd(inp, p[b + "/expansion_1x1_conv2d/conv2d/kernel"].reshape(cin, cin * exp), b + "/expansion_1x1_conv2d", ACT_RELU6, tape)
      wd = p[b + "/depthwise_conv2d/SeparableConv2d/depthwise_weights"].reshape(21, cin * exp)
      yd = self._dw(a, wd, H, W)
      mean, rstd, shift = self._bn_fwd(yd, b + "/depthwise_conv2d")
      ad = self._act(yd, rstd, shift, ACT_RELU6)
      tape.append(("dw", a, wd, yd, mean, rstd, shift, b, H, W))
      out = self._cba_fwd(ad, p[b + "/projection_1x1_conv2d/conv2d/kernel"].reshape(cin * exp, cout), b + "/projection_1x1_conv2d", ACT_NONE, tape)
      if cout != cin:
        sc = self._cba_fwd(inp, p[b + "/1x1_conv2d/conv2d/kernel"].reshape(cin, cout), b + "/1x1_conv2d", ACT_NONE, tape)
        tape.append(("add_sc",))
        net = out + sc
      else:
        tape.append(("add_id",))
        net = out + inp
      if pool:
        pooled, Ho, Wn = self._pool(net, H, W, (2, 2), (1, 2))
        tape.append(("pool", net, H, W, (2, 2), (1, 2)))
        net, W = pooled, Wn
    s8 = PREFIX + "block8_0/conv2d"
    feat = self._cba_fwd(net, p[s8 + "/conv2d/kernel"].reshape(256, 256), s8, ACT_RELU, tape)
    enc_in, Ho, Wn = self._pool(feat, H, W, (5, 3), (5, 3))
    assert Ho == T and Wn == 1, (Ho, Wn)
    tape.append(("pool", feat, H, W, (5, 3), (5, 3)))

    def dense(x, wname, bname, act, mask):
      z = torch.addmm(p[bname], x, p[wname])
      y = z if (act == ACT_NONE and mask is None) else self._act(z, None, None, act, mask)
      tape.append(("dense", x, wname, bname, y, act, mask))
      return y
    e = dense(enc_in, "mfcc_encoder/dense/kernel", "mfcc_encoder/dense/bias", ACT_LRELU, mk("enc", 256))
    c1 = dense(e, "rnn_module/dense/kernel", "rnn_module/dense/bias", ACT_LRELU, None)
    wg, wc = p[GRU + "gates/kernel"], p[GRU + "candidate/kernel"]
    xg = torch.addmm(p[GRU + "gates/bias"], c1, wg[:256])
    xc = torch.addmm(p[GRU + "candidate/bias"], c1, wc[:256])
    whg, whc = wg[256:].contiguous(), wc[256:].contiguous()
    rnn, sr, su, scand, shp = (torch.empty(B * T, 256, dtype=torch.float32, device=self.dev) for _ in range(5))
    _lib.check(L.vp_gru_train_fwd(_ptr(xg), _ptr(xc), _ptr(whg), _ptr(whc), _ptr(seq), _ptr(rnn), _ptr(sr), _ptr(su), _ptr(scand), _ptr(shp), B, T,
                                  _stream()), "vp_gru_train_fwd")
    mr = mk("rnn", 256)
    rnn_m = rnn if mr is None else self._act(rnn, None, None, ACT_NONE, mr)
    d0 = dense(rnn_m, "bfm_coeff_decoder/dense/kernel", "bfm_coeff_decoder/dense/bias", ACT_LRELU, mk("d0", 128))
    d1 = dense(d0, "bfm_coeff_decoder/dense_1/kernel", "bfm_coeff_decoder/dense_1/bias", ACT_LRELU, mk("d1", 64))
    o = dense(d1, "bfm_coeff_decoder/dense_2/kernel", "bfm_coeff_decoder/dense_2/bias", ACT_NONE, None)
    o = o.clone()
    o[:, 16:20] += (ears.reshape(B * T, 1) * self.ears_scale)                       # + tf.pad(ears * [-2,-2,-2,-4], [16, 44])
    self.last_out = o.view(B, T, 64)

    loss_data, do = self._vertex_loss(o, bfm_coeffs, seq)

    # ---- backward ----------------------------------------------------------------------------------------------------------------
    self.grads.zero_()

    def dense_bwd(dy):
      _, x, wname, bname, y, act, mask = tape.pop()
      dz = dy if (act == ACT_NONE and mask is None) else self._act_bwd(dy, y, act, mask)
      torch.mm(x.t(), dz, out=g[wname])
      torch.sum(dz, 0, out=g[bname])
      return torch.mm(dz, p[wname].t())
    d = dense_bwd(do)
    d = dense_bwd(d)
    d = dense_bwd(d)                                                                 # d loss / d rnn_m
    if mr is not None:
      d = d * mr
    dag, dac = torch.empty(B * T, 512, dtype=torch.float32, device=self.dev), torch.empty(B * T, 256, dtype=torch.float32, device=self.dev)
    _lib.check(L.vp_gru_train_bwd(_ptr(d.contiguous()), _ptr(whg), _ptr(whc), _ptr(seq), _ptr(sr), _ptr(su), _ptr(scand), _ptr(shp), _ptr(dag), _ptr(dac),
                                  B, T, _stream()), "vp_gru_train_bwd")
    torch.mm(c1.t(), dag, out=g[GRU + "gates/kernel"][:256])
    torch.mm(shp.t(), dag, out=g[GRU + "gates/kernel"][256:])
    torch.sum(dag, 0, out=g[GRU + "gates/bias"])
    torch.mm(c1.t(), dac, out=g[GRU + "candidate/kernel"][:256])
    torch.mm((sr * shp).t(), dac, out=g[GRU + "candidate/kernel"][256:])
    torch.sum(dac, 0, out=g[GRU + "candidate/bias"])
    d = torch.mm(dag, wg[:256].t()) + torch.mm(dac, wc[:256].t())                      # d loss / d c1
    d = dense_bwd(d)
    d = dense_bwd(d)                                                                 # d loss / d enc_in  [B*T, 256]

    def cba_bwd(da, wgrad=True):
      """-> d loss / d x (None for the stem)"""
      _, x, kernel, y, mean, rstd, shift, act, scope = tape.pop()
      dy = self._bn_act_bwd(da, y, mean, rstd, shift, act, scope)
      if x.shape[1] == 48:                                                           # stem: no input gradient
        g[PREFIX + "block0_0/conv2d/conv2d/kernel"].view(45, 32).copy_(torch.mm(x.t(), dy)[:45])
        return None
      torch.mm(x.t(), dy, out=self._grad2d(kernel))
      return torch.mm(dy, kernel.t())
    while tape:
      kind = tape[-1][0]
      if kind == "pool":
        _, x, h_, w_, k_, s_ = tape.pop()
        d = self._pool_bwd(x, d, h_, w_, k_, s_)
        H, W = h_, w_
      elif kind in ("add_sc", "add_id"):
        tape.pop()
        dsum = d
        dsc = cba_bwd(dsum) if kind == "add_sc" else dsum                            # shortcut conv (its tape entry lies on top) / identity
        dad = cba_bwd(dsum)                                                          # projection conv
        _, a_in, wd, yd, mean, rstd, shift, b, h_, w_ = tape.pop()
        dyd = self._bn_act_bwd(dad, yd, mean, rstd, shift, ACT_RELU6, b + "/depthwise_conv2d")
        self._dw_wgrad(a_in, dyd, h_, w_, g[b + "/depthwise_conv2d/SeparableConv2d/depthwise_weights"])
        da = self._dw(dyd, wd.flip(0).contiguous(), h_, w_)
        d = cba_bwd(da)                                                              # expansion conv
        d.add_(dsc)
      elif kind == "cba":                                                            # block8_0 (1x1) or the stem
        d = cba_bwd(d)
      else:
        raise AssertionError(kind)

    # ---- regulariser, clip_by_global_norm, Adam, moving averages ------------------------------------------------------------------------
    part = torch.empty(L.vp_sumsq_partials(self.ntrain), dtype=torch.float64, device=self.dev)
    _lib.check(L.vp_l2_regulariser(_ptr(self.arena), _ptr(self.l2mask), _ptr(self.grads), self.ntrain, L2_SCALE, _ptr(part), _stream()), "vp_l2_regulariser")
    loss = loss_data + 0.5 * L2_SCALE * part.sum()
    ss = self._sumsq(self.grads)
    self._last_ss = ss
    if apply:
      _lib.check(L.vp_adam_tf_clipped(_ptr(self.arena), _ptr(self.grads), _ptr(self.m), _ptr(self.v), self.ntrain, _ptr(self.lr_t), _ptr(ss), self.clip,
                                      0.9, 0.999, 1e-8, _stream()), "vp_adam_tf_clipped")
      _lib.check(L.vp_moving_update(_ptr(self.arena[self.ntrain:]), _ptr(self.bstats), _ptr(self._moving_factor()), self.bstats.numel(), BN_DECAY,
                                    _stream()), "vp_moving_update")
    else:
      gn = torch.sqrt(ss)
      self.grads.mul_((self.clip / torch.clamp(gn, min=self.clip)).to(torch.float32))
    return torch.stack([loss, loss_data, torch.sqrt(ss)])

  def _vertex_loss(self, o, bfm_coeffs, seq):
    """add_cost_function (bfmnet.py:229-271): both face shapes share the identity coefficients, so their difference is
    exBase . (ex_true - ex_pred).  o [B*T,64] -> (data loss as a float64 device scalar, d loss / d o [B*T,64])."""
    B, T, L = self.B, self.T, self.L
    delta = bfm_coeffs.reshape(B * T, -1)[:, 80:144] - o
    D = torch.mm(delta, self.exbase.t())                                             # [B*T, 3n]
    gD = torch.empty_like(D)
    npart = L.vp_vertex_loss_partials(B, self.J)
    part = torch.empty(npart, dtype=torch.float64, device=self.dev)
    _lib.check(L.vp_bfm_vertex_loss(_ptr(D), _ptr(self.vmask), _ptr(seq), B, T, self.J, _ptr(gD), _ptr(part), _stream()), "vp_bfm_vertex_loss")
    return part.sum(), -torch.mm(gD, self.exbase)

  def regulariser(self):
    reg = torch.zeros((), dtype=torch.float64, device=self.dev)
    for n in self.p:
      if regularised(n):
        reg = reg + self._sumsq(self.p[n])
    return 0.5 * L2_SCALE * reg

  def eval_loss(self, coeff, bfm_coeffs, seq_len):
    """The Loss node of build_eval_op (bfmnet.py:273-289): the same cost on coefficients predicted in inference mode."""
    seq = torch.as_tensor(seq_len, dtype=torch.int32, device=self.dev).contiguous()
    o = coeff.to(self.dev, torch.float32).reshape(self.B * self.T, 64).contiguous()
    ld, _ = self._vertex_loss(o, bfm_coeffs.to(self.dev, torch.float32), seq)
    return float(ld + self.regulariser())

  def draw_masks(self, drop_rate, inner_rate=0.25, generator=None):
    """One draw of the four dropout masks: keep with probability 1-rate, kept entries scaled by 1/(1-rate).  `drop_rate` is
    params.training['drop_rate'] and reaches only the tf.layers.dropout after the encoder's dense layer (bfmnet.py:199); RNNModule's
    DropoutWrapper and BFMCoeffDecoder's two tf.nn.dropout keep their constructor default 0.25 (bfmnet.py:45,77,209).  The draws come
    from torch's device generator, not TensorFlow's stream."""
    def mk(c, rate):
      if not rate:
        return None
      keep = 1.0 - float(rate)
      return (torch.rand(self.B, self.T, c, device=self.dev, generator=generator) < keep).to(torch.float32) / keep
    return {"enc": mk(256, drop_rate), "rnn": mk(256, inner_rate), "d0": mk(128, inner_rate), "d1": mk(64, inner_rate)}

