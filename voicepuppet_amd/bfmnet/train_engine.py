"""BFMNet.build_train_op on the device (reference: voicepuppet/bfmnet/bfmnet.py:215-323 over tinynet.py:7-212).

One `train_step` = training-mode forward (batch-statistics batch_norm, dropout masks), the vertex-space loss of add_cost_function
plus the l2 regulariser, the full backward pass, clip_by_global_norm and one tf.train.AdamOptimizer update, plus the moving-average
update of every batch_norm.  Division of labour: everything that is a plain matrix product (1x1 convolutions and dense layers on
[pixels, channels] matrices, the GRU's input / recurrent weight gradients, the [B*T, 64] x [64, 3n] face-shape products) goes
through vp_mm_fwd_f32 / vp_mm_bwd_data_f32 / vp_mm_bwd_weight_f32 (csrc/mm_api.hip: the repo's own float32-MFMA implicit-GEMM and
weight-gradient kernels, the same ones the PixReferNet float32 path runs on; no vendor GEMM library); everything else -
batch-norm statistics and backward, activations and dropout, the depthwise 7x3
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


class BFMNetTrainEngine:
  def __init__(self, batch, frames, model, lr=1e-4, max_grad_norm=50.0, num_mel_bins=80, side_stream=True):
    """model: dict with exBase [3n,64] and vmask [3n] (the mouth-weighted vertex mask of bfmnet.py:131-134); idBase / meanshape cancel in
    every term of the loss (both face shapes share the identity coefficients) and are not needed on the device."""
    if not torch.cuda.is_available():
      raise RuntimeError("BFMNetTrainEngine needs an MI355X (no CPU fallback)")
    self.L = _lib.lib()
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
    self._gdpad = None
    self._pk, self._pk_pending, self._pk_ready, self._pk_arena, self._packed = {}, {}, False, (None, 0), None
    self._graphs = {}
    # weight gradients (1x1 / depthwise / dense / GRU) run on a second stream beside the data-gradient chain (side_stream=False: one stream)
    self._side = torch.cuda.Stream() if side_stream else None
    self._hold, self._mmkey = [], "mm"

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

  def _act(self, x, scale, shift, act, mask=None, add=None):
    """act(scale * x + shift) * mask (+ add: the residual branch of a block, fused into the same pass)"""
    P, C = x.shape
    y = torch.empty_like(x)
    if add is not None:
      _lib.check(self.L.vp_affine_act_add_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(mask), _ptr(add), P, C, act, _ptr(y), _stream()), "vp_affine_act_add_fwd")
    else:
      _lib.check(self.L.vp_affine_act_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(mask), P, C, act, _ptr(y), _stream()), "vp_affine_act_fwd")
    return y

  def _act_bwd(self, dy, ya, act, mask=None):
    dx = torch.empty_like(dy)
    _lib.check(self.L.vp_act_bwd(_ptr(dy), _ptr(ya), _ptr(mask), dy.numel(), act, _ptr(dx), _stream()), "vp_act_bwd")
    return dx

  def _dw(self, x2d, w21, H, W, backward=False):
    """depthwise 7x3; backward=True: the data gradient (the same taps walked in reverse)"""
    C = x2d.shape[1]
    y = torch.empty_like(x2d)
    fn = self.L.vp_dwconv7x3_bwd_data if backward else self.L.vp_dwconv7x3_raw
    _lib.check(fn(_ptr(x2d), _ptr(w21), _ptr(y), self.B, H, W, C, _stream()), "vp_dwconv7x3")
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

  # ---- matrix products: the repo's own float32 MFMA kernels behind vp_mm_* (include/vp_hip.h), no vendor GEMM library --------------------
  def _mm_ws(self, P, K, N):
    """One workspace per stream for every product of the step (they run back to back on that stream); its first 256 bytes are the
    kernels' zero page (include/vp_hip.h): allocated zeroed, never written."""
    need = self.L.vp_mm_workspace_bytes(int(P), int(K), int(N))
    w = self._ws.get(self._mmkey)
    if w is None or w.numel() < need:
      if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("the matrix-product workspace must exist before a graph capture (run one eager step first)")
      w = self._ws[self._mmkey] = torch.zeros(int(need), dtype=torch.uint8, device=self.dev)
    return w

  def _fork(self, fn, *keep):
    """Run fn() - weight-gradient launches, nothing the data-gradient chain waits for - on the side stream, ordered behind everything
    enqueued on the current stream so far.  The tensors it reads are held until _join(): the caching allocator must not hand their memory
    to a later allocation of the main stream while the side stream still reads it (also inside a graph capture's private pool)."""
    if self._side is None or torch.cuda.is_current_stream_capturing():
      fn()            # (a captured graph replays one stream: hipGraph branches measured slower than the chain, 8.97 vs 8.49 ms at batch 4)
      return
    self._side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(self._side):
      self._mmkey = "mm_side"
      try:
        fn()
      finally:
        self._mmkey = "mm"
    self._hold.extend(keep)

  def _join(self):
    if self._side is not None:
      torch.cuda.current_stream().wait_stream(self._side)
    self._hold = []

  # Weight matrices are packed into the kernels' chunk-major layout AHEAD of the products: the first (eager) step registers every
  # (matrix, direction, shape) it multiplies with; from then on ONE launch at the top of a step packs all arena matrices from the
  # freshly updated parameters (vp_mm_pack_table), constant matrices (exBase) were packed once, and the products read the packed blocks.
  def _packed_for(self, w, w_t, P, K, N, direction):
    """-> device pointer of the packed block of `w` for this product, or None while the table does not hold it yet."""
    a0, esz = self.arena.data_ptr(), 4
    in_arena = a0 <= w.data_ptr() < a0 + self.arena.numel() * esz
    if not in_arena and w.data_ptr() != self.exbase.data_ptr():
      return None                                      # a temporary (the zero-padded stem kernel): packed per product
    key = ((w.data_ptr() - a0) // esz if in_arena else -w.data_ptr(), w.stride(0), bool(w_t), int(P), int(K), int(N), direction)
    hit = self._pk.get(key)
    if hit is not None:
      return hit
    if not self._pk_ready and key not in self._pk_pending:
      self._pk_pending[key] = (w if not in_arena else None)
    return None

  def _build_pack_table(self):
    L = self.L
    dsz = int(L.vp_mm_pack_desc_bytes())
    groups = {"arena": [], "const": []}
    for key, wconst in self._pk_pending.items():
      groups["const" if wconst is not None else "arena"].append((key, wconst))
    total = sum(int(L.vp_mm_packed_bytes(k[3], k[4], k[5], k[6])) for k in self._pk_pending)
    self._packed = torch.zeros(max(total, 256), dtype=torch.uint8, device=self.dev)
    off = 0
    for name, items in groups.items():
      host = ctypes.create_string_buffer(dsz * max(len(items), 1))
      for i, (key, wconst) in enumerate(items):
        w_off, ldw, w_t, P, K, N, direction = key
        src = 0 if wconst is not None else w_off
        # one table per master pointer: a constant matrix is its own master (offset 0)
        _lib.check(L.vp_mm_pack_desc(src, ldw, 1 if w_t else 0, P, K, N, direction, off // 4, ctypes.byref(host, i * dsz)), "vp_mm_pack_desc")
        self._pk[key] = ctypes.c_void_p(self._packed.data_ptr() + off)
        off += int(L.vp_mm_packed_bytes(P, K, N, direction))
      dev = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.dev)
      if name == "arena":
        self._pk_arena = (dev, len(items))
      else:
        # constants: pack now, once (each against its own base pointer: one launch per matrix)
        for i, (key, wconst) in enumerate(items):
          one = dev[i * dsz:(i + 1) * dsz].clone()
          _lib.check(L.vp_mm_pack_table(_ptr(one), 1, _ptr(wconst), _ptr(self._packed), _stream()), "vp_mm_pack_table")
          self._keep_const = getattr(self, "_keep_const", []) + [one]
    self._pk_pending = {}
    self._pk_ready = True

  def _pack_weights(self):
    """One launch: every arena matrix of the step, both layouts, from the current parameters."""
    if self._pk_ready and self._pk_arena[1] > 0:
      dev, n = self._pk_arena
      _lib.check(self.L.vp_mm_pack_table(_ptr(dev), n, _ptr(self.arena), _ptr(self._packed), _stream()), "vp_mm_pack_table")

  def _mm(self, x, w, bias=None, w_t=False, out=None):
    """x [P, K] . w [K, N] (+ bias) -> [P, N];  w_t: w is stored [N, K].  Row strides are taken from the tensors."""
    P, K = x.shape
    N = w.shape[0] if w_t else w.shape[1]
    assert x.stride(1) == 1 and w.stride(1) == 1 and K % 16 == 0, (x.shape, w.shape)
    y = out if out is not None else torch.empty(P, N, dtype=torch.float32, device=self.dev)
    pk = self._packed_for(w, w_t, P, K, N, 0)
    if pk is not None:
      _lib.check(self.L.vp_mm_fwd_f32_packed(_ptr(x), x.stride(0), pk, _ptr(bias), _ptr(y), y.stride(0), P, K, N, _ptr(self._mm_ws(P, K, N)), _stream()),
                 "vp_mm_fwd_f32_packed")
    else:
      _lib.check(self.L.vp_mm_fwd_f32(_ptr(x), x.stride(0), _ptr(w), w.stride(0), 1 if w_t else 0, _ptr(bias), _ptr(y), y.stride(0), P, K, N,
                                      _ptr(self._mm_ws(P, K, N)), _stream()), "vp_mm_fwd_f32")
    return y

  def _mm_dx(self, dy, w, w_t=False, out=None, accumulate=False, n=None):
    """dy [P, N] . w [K, N]^T -> [P, K] (optionally added to `out`);  w_t: w is stored [N, K];  n: real columns of dy when it is
    stored with zero padding up to a multiple of 16."""
    P = dy.shape[0]
    N = int(n) if n is not None else dy.shape[1]
    K = w.shape[1] if w_t else w.shape[0]
    assert dy.stride(1) == 1 and w.stride(1) == 1 and dy.stride(0) >= -(-N // 16) * 16, (dy.shape, dy.stride(), N)
    dx = out if out is not None else torch.empty(P, K, dtype=torch.float32, device=self.dev)
    pk = self._packed_for(w, w_t, P, K, N, 1)
    if pk is not None:
      _lib.check(self.L.vp_mm_bwd_data_f32_packed(_ptr(dy), dy.stride(0), pk, _ptr(dx), dx.stride(0), 1 if accumulate else 0, P, K, N,
                                                  _ptr(self._mm_ws(P, K, N)), _stream()), "vp_mm_bwd_data_f32_packed")
    else:
      _lib.check(self.L.vp_mm_bwd_data_f32(_ptr(dy), dy.stride(0), _ptr(w), w.stride(0), 1 if w_t else 0, _ptr(dx), dx.stride(0), 1 if accumulate else 0,
                                           P, K, N, _ptr(self._mm_ws(P, K, N)), _stream()), "vp_mm_bwd_data_f32")
    return dx

  def _colsum(self, x, out):
    """out[c] = sum over rows of x: the bias gradients (vp_colsum_f32; it was torch.sum, an at::native reduction)"""
    assert x.is_contiguous() and out.is_contiguous() and out.numel() == x.shape[1]
    _lib.check(self.L.vp_colsum_f32(_ptr(x), int(x.shape[0]), int(x.shape[1]), _ptr(out), _stream()), "vp_colsum_f32")

  def _mul(self, a, b):
    out = torch.empty_like(a)
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    _lib.check(self.L.vp_mul_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "vp_mul_f32")
    return out

  def _mm_dw(self, x, dy, out, k_real=None):
    """x [P, K]^T . dy [P, N] -> out [k_real, N] (contiguous; k_real < K: x carries zero padding columns, the stem)"""
    P, K = x.shape
    N = dy.shape[1]
    kr = K if k_real is None else int(k_real)
    assert x.stride(1) == 1 and dy.stride(1) == 1 and out.is_contiguous() and out.numel() == kr * N, (x.shape, dy.shape, out.shape)
    _lib.check(self.L.vp_mm_bwd_weight_f32(_ptr(x), x.stride(0), _ptr(dy), dy.stride(0), _ptr(out), P, K, kr, N, _ptr(self._mm_ws(P, K, N)), _stream()),
               "vp_mm_bwd_weight_f32")
    return out

  def _sumsq(self, x):
    n = self.L.vp_sumsq_partials(x.numel())
    part = torch.empty(n, dtype=torch.float64, device=self.dev)
    _lib.check(self.L.vp_sumsq(_ptr(x), x.numel(), _ptr(part), _stream()), "vp_sumsq")
    return self._sum64(part)

  def _sum64(self, part, scale=1.0, add=None):
    """Float64 device scalar = (add or 0) + scale * sum(part), by the library's one-block kernel (no framework reduction on the path)."""
    out = torch.empty((), dtype=torch.float64, device=self.dev)
    _lib.check(self.L.vp_sum_f64(_ptr(part), part.numel(), float(scale), _ptr(add) if add is not None else None, _ptr(out), _stream()), "vp_sum_f64")
    return out

  # ---- conv + batch-norm + activation, forward / backward -----------------------------------------------------------------------
  def _cba_fwd(self, x, kernel, bn_scope, act, tape, add=None):
    """x [P, cin] . kernel [cin, cout] -> batch_norm -> act (+ add).  tape gets what the backward needs."""
    y = self._mm(x, kernel)
    mean, rstd, shift = self._bn_fwd(y, bn_scope)
    a = self._act(y, rstd, shift, act, add=add)
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

  def train_step_auto(self, ears, mfccs, bfm_coeffs, seq_len, drop_rate=0.25, inner_rate=0.25):
    """The faster of the two schedules for this batch size: up to 8 clips the step is a chain of launch-latency-bound kernels and the
    hipGraph replay wins (8.5 vs 9.2 ms eager at batch 4); above that the eager step with the weight gradients on the second stream
    does (24.3 vs 26.3 ms at batch 32: the graph replays one stream)."""
    if self.B <= 8 or self._side is None:
      return self.train_step_graphed(ears, mfccs, bfm_coeffs, seq_len, drop_rate, inner_rate)
    return self.train_step(ears, mfccs, bfm_coeffs, seq_len, masks=self.draw_masks(drop_rate, inner_rate))

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
    self._pack_weights()
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
      # the block's sum out + shortcut rides on the last normalisation pass (the shortcut conv of a widening block runs after the
      # projection conv: the reference's creation order of the variables, and the order the tape is unwound in)
      pk, ps = p[b + "/projection_1x1_conv2d/conv2d/kernel"].reshape(cin * exp, cout), b + "/projection_1x1_conv2d"
      if cout != cin:
        out = self._cba_fwd(ad, pk, ps, ACT_NONE, tape)
        net = self._cba_fwd(inp, p[b + "/1x1_conv2d/conv2d/kernel"].reshape(cin, cout), b + "/1x1_conv2d", ACT_NONE, tape, add=out)
        tape.append(("add_sc",))
      else:
        net = self._cba_fwd(ad, pk, ps, ACT_NONE, tape, add=inp)
        tape.append(("add_id",))
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
      z = self._mm(x, p[wname], p[bname])
      y = z if (act == ACT_NONE and mask is None) else self._act(z, None, None, act, mask)
      tape.append(("dense", x, wname, bname, y, act, mask))
      return y
    e = dense(enc_in, "mfcc_encoder/dense/kernel", "mfcc_encoder/dense/bias", ACT_LRELU, mk("enc", 256))
    c1 = dense(e, "rnn_module/dense/kernel", "rnn_module/dense/bias", ACT_LRELU, None)
    wg, wc = p[GRU + "gates/kernel"], p[GRU + "candidate/kernel"]
    xg = self._mm(c1, wg[:256], p[GRU + "gates/bias"])
    xc = self._mm(c1, wc[:256], p[GRU + "candidate/bias"])
    # the recurrent halves of the two GRUCell kernels, as they are (forward) and transposed (backward): one launch, no at::native copies
    whg, whg_t = (torch.empty(256, 512, dtype=torch.float32, device=self.dev), torch.empty(512, 256, dtype=torch.float32, device=self.dev))
    whc, whc_t = (torch.empty(256, 256, dtype=torch.float32, device=self.dev), torch.empty(256, 256, dtype=torch.float32, device=self.dev))
    _lib.check(L.vp_gru_split_recurrent(_ptr(wg), _ptr(wc), _ptr(whg), _ptr(whc), _ptr(whg_t), _ptr(whc_t), _stream()), "vp_gru_split_recurrent")
    rnn, sr, su, scand, shp = (torch.empty(B * T, 256, dtype=torch.float32, device=self.dev) for _ in range(5))
    _lib.check(L.vp_gru_train_fwd(_ptr(xg), _ptr(xc), _ptr(whg), _ptr(whc), _ptr(seq), _ptr(rnn), _ptr(sr), _ptr(su), _ptr(scand), _ptr(shp), B, T,
                                  _stream()), "vp_gru_train_fwd")
    mr = mk("rnn", 256)
    rnn_m = rnn if mr is None else self._act(rnn, None, None, ACT_NONE, mr)
    d0 = dense(rnn_m, "bfm_coeff_decoder/dense/kernel", "bfm_coeff_decoder/dense/bias", ACT_LRELU, mk("d0", 128))
    d1 = dense(d0, "bfm_coeff_decoder/dense_1/kernel", "bfm_coeff_decoder/dense_1/bias", ACT_LRELU, mk("d1", 64))
    o = dense(d1, "bfm_coeff_decoder/dense_2/kernel", "bfm_coeff_decoder/dense_2/bias", ACT_NONE, None)
    # + tf.pad(ears * [-2,-2,-2,-4], [16, 44]), in place (the last dense layer has no activation: its backward never reads its output)
    _lib.check(L.vp_add_ears_f32(_ptr(o), _ptr(ears.reshape(B * T).contiguous()), B * T, _stream()), "vp_add_ears_f32")
    self.last_out = o.view(B, T, 64)

    loss_data, do = self._vertex_loss(o, bfm_coeffs, seq)

    # ---- backward ----------------------------------------------------------------------------------------------------------------
    self.grads.zero_()

    def dense_bwd(dy):
      _, x, wname, bname, y, act, mask = tape.pop()
      dz = dy if (act == ACT_NONE and mask is None) else self._act_bwd(dy, y, act, mask)
      def wg(x=x, dz=dz, wname=wname, bname=bname):
        self._mm_dw(x, dz, g[wname])
        self._colsum(dz, g[bname])
      self._fork(wg, x, dz)
      return self._mm_dx(dz, p[wname])
    d = dense_bwd(do)
    d = dense_bwd(d)
    d = dense_bwd(d)                                                                 # d loss / d rnn_m
    if mr is not None:
      d = self._mul(d, mr)
    dag, dac = torch.empty(B * T, 512, dtype=torch.float32, device=self.dev), torch.empty(B * T, 256, dtype=torch.float32, device=self.dev)
    _lib.check(L.vp_gru_train_bwd(_ptr(d.contiguous()), _ptr(whg_t), _ptr(whc_t), _ptr(seq), _ptr(sr), _ptr(su), _ptr(scand), _ptr(shp), _ptr(dag), _ptr(dac),
                                  B, T, _stream()), "vp_gru_train_bwd")
    srh = self._mul(sr, shp)
    def gru_wg():
      self._mm_dw(c1, dag, g[GRU + "gates/kernel"][:256])
      self._mm_dw(shp, dag, g[GRU + "gates/kernel"][256:])
      self._colsum(dag, g[GRU + "gates/bias"])
      self._mm_dw(c1, dac, g[GRU + "candidate/kernel"][:256])
      self._mm_dw(srh, dac, g[GRU + "candidate/kernel"][256:])
      self._colsum(dac, g[GRU + "candidate/bias"])
    self._fork(gru_wg, c1, shp, srh, dag, dac)
    d = self._mm_dx(dag, wg[:256])
    self._mm_dx(dac, wc[:256], out=d, accumulate=True)                               # d loss / d c1
    d = dense_bwd(d)
    d = dense_bwd(d)                                                                 # d loss / d enc_in  [B*T, 256]

    def cba_bwd(da, into=None):
      """-> d loss / d x (None for the stem); into: a gradient of the same tensor the result is ADDED to, in place (the shortcut's)"""
      _, x, kernel, y, mean, rstd, shift, act, scope = tape.pop()
      dy = self._bn_act_bwd(da, y, mean, rstd, shift, act, scope)
      if x.shape[1] == 48:                                                           # stem: no input gradient
        self._fork(lambda: self._mm_dw(x, dy, g[PREFIX + "block0_0/conv2d/conv2d/kernel"].view(45, 32), k_real=45), x, dy)
        return None
      gk = self._grad2d(kernel)
      self._fork(lambda: self._mm_dw(x, dy, gk), x, dy)
      if into is not None:
        return self._mm_dx(dy, kernel, out=into, accumulate=True)
      return self._mm_dx(dy, kernel)
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
        gdw = g[b + "/depthwise_conv2d/SeparableConv2d/depthwise_weights"]
        self._fork(lambda a_in=a_in, dyd=dyd, h_=h_, w_=w_, gdw=gdw: self._dw_wgrad(a_in, dyd, h_, w_, gdw), a_in, dyd)
        da = self._dw(dyd, wd, h_, w_, backward=True)
        d = cba_bwd(da, into=dsc)                                                    # expansion conv, added to the shortcut's gradient
      elif kind == "cba":                                                            # block8_0 (1x1) or the stem
        d = cba_bwd(d)
      else:
        raise AssertionError(kind)

    self._join()                                                                     # every weight gradient is in the arena
    # ---- regulariser, clip_by_global_norm, Adam, moving averages ------------------------------------------------------------------------
    part = torch.empty(L.vp_sumsq_partials(self.ntrain), dtype=torch.float64, device=self.dev)
    _lib.check(L.vp_l2_regulariser(_ptr(self.arena), _ptr(self.l2mask), _ptr(self.grads), self.ntrain, L2_SCALE, _ptr(part), _stream()), "vp_l2_regulariser")
    reg = self._sum64(part)
    ss = self._sumsq(self.grads)
    self._last_ss = ss
    if apply:
      _lib.check(L.vp_adam_tf_clipped(_ptr(self.arena), _ptr(self.grads), _ptr(self.m), _ptr(self.v), self.ntrain, _ptr(self.lr_t), _ptr(ss), self.clip,
                                      0.9, 0.999, 1e-8, _stream()), "vp_adam_tf_clipped")
      _lib.check(L.vp_moving_update(_ptr(self.arena[self.ntrain:]), _ptr(self.bstats), _ptr(self._moving_factor()), self.bstats.numel(), BN_DECAY,
                                    _stream()), "vp_moving_update")
    else:
      _lib.check(L.vp_clip_scale_f32(_ptr(self.grads), self.ntrain, _ptr(ss), self.clip, _stream()), "vp_clip_scale_f32")
    if self._pk_pending and not torch.cuda.is_current_stream_capturing():
      self._build_pack_table()                                                       # after the first eager step: the products are known
    rep = torch.empty(3, dtype=torch.float64, device=self.dev)
    _lib.check(L.vp_bfm_step_report(_ptr(loss_data), _ptr(reg), 0.5 * L2_SCALE, _ptr(ss), _ptr(rep), _stream()), "vp_bfm_step_report")
    return rep

  def _vertex_loss(self, o, bfm_coeffs, seq):
    """add_cost_function (bfmnet.py:229-271): both face shapes share the identity coefficients, so their difference is
    exBase . (ex_true - ex_pred).  o [B*T,64] -> (data loss as a float64 device scalar, d loss / d o [B*T,64])."""
    B, T, L = self.B, self.T, self.L
    J = self.J
    delta = (bfm_coeffs.reshape(B * T, -1)[:, 80:144] - o).contiguous()
    D = self._mm(delta, self.exbase, w_t=True)                                       # delta . exBase^T  [B*T, 3n] (exBase is stored [3n, 64])
    gD = torch.empty_like(D)
    npart = L.vp_vertex_loss_partials(B, J)
    part = torch.empty(npart, dtype=torch.float64, device=self.dev)
    _lib.check(L.vp_bfm_vertex_loss(_ptr(D), _ptr(self.vmask), _ptr(seq), B, T, J, _ptr(gD), _ptr(part), _stream()), "vp_bfm_vertex_loss")
    # gD . exBase contracts over the 3n vertex coordinates (not a multiple of the kernels' 16-float K chunk): a zero-padded copy
    gp = self._gdpad
    if gp is None or gp.shape[0] != B * T:
      gp = self._gdpad = torch.zeros(B * T, -(-J // 16) * 16, dtype=torch.float32, device=self.dev)
    gp[:, :J].copy_(gD)
    return self._sum64(part), -self._mm_dx(gp, self.exbase, w_t=True, n=J)

  def regulariser(self):
    reg = None
    for n in self.p:
      if regularised(n):
        part = torch.empty(self.L.vp_sumsq_partials(self.p[n].numel()), dtype=torch.float64, device=self.dev)
        _lib.check(self.L.vp_sumsq(_ptr(self.p[n]), self.p[n].numel(), _ptr(part), _stream()), "vp_sumsq")
        reg = self._sum64(part, add=reg)
    return float(reg) * 0.5 * L2_SCALE if reg is not None else 0.0

  def eval_loss(self, coeff, bfm_coeffs, seq_len):
    """The Loss node of build_eval_op (bfmnet.py:273-289): the same cost on coefficients predicted in inference mode."""
    seq = torch.as_tensor(seq_len, dtype=torch.int32, device=self.dev).contiguous()
    o = coeff.to(self.dev, torch.float32).reshape(self.B * self.T, 64).contiguous()
    ld, _ = self._vertex_loss(o, bfm_coeffs.to(self.dev, torch.float32), seq)
    return float(ld) + self.regulariser()

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

