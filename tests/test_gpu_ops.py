"""-m gpu: single-op parity of the HIP kernels against the numpy oracle (oracle/nn_ops.py),
called through the C ABI.  Tolerances: f32 path 1e-5 relative L2 (f32 MFMA == fmaf chain);
bf16 path 1e-2 on bf16-rounded operands (bf16 storage of the result, f32 accumulate)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nn_ops as ops
from voicepuppet_amd import _lib

import gpu_util as gu

pytestmark = pytest.mark.gpu
TOL = {"f32": 2e-5, "bf16": 1e-2}
ACTS = {0: lambda v: v, 1: lambda v: ops.lrelu(v, 0.2), 2: ops.relu, 3: np.tanh, 4: ops.sigmoid}

# (kind, n, h, w, cin, cout, k, s, p, in_act, out_act, affine)
FWD_CASES = [
    (0, 2, 16, 16, 8, 64, 4, 2, 1, 0, 0, False),      # encoder_1-like (padded thin input)
    (0, 2, 16, 16, 64, 128, 4, 2, 1, 1, 0, True),     # encoder_k: lrelu(bn(x)) -> conv
    (0, 3, 9, 9, 32, 64, 4, 1, 1, 1, 0, True),        # D layer_4-like: stride 1, odd size
    (0, 3, 9, 9, 64, 1, 4, 1, 1, 1, 0, True),         # D layer_5-like: one output channel
    (0, 2, 12, 12, 8, 64, 3, 1, 1, 0, 2, False),      # VGG conv1_1-like (K = 72 is padded)
    (0, 3, 16, 32, 8, 64, 3, 1, 1, 0, 2, False),      # same with power-of-two sides: the direct 8-channel kernel (3 MFMA steps)
    (0, 2, 12, 12, 64, 64, 3, 1, 1, 0, 2, False),     # VGG conv + relu
    (0, 2, 32, 48, 64, 64, 3, 1, 1, 0, 2, False),     # VGG conv1_2 geometry (sides multiples of 8 x 16): register-resident weights (conv_c64.hip); bwd-data: its flipped form
    (0, 1, 8, 16, 64, 64, 3, 1, 1, 0, 0, False),      # ... two tiles: almost every patch row / column outside the image is padding
    (0, 2, 32, 48, 64, 128, 3, 1, 1, 0, 2, False),    # VGG conv2_1 geometry: the four-wave form of the same kernel (128 output channels)
    (0, 2, 32, 48, 128, 128, 3, 1, 1, 0, 2, False),   # VGG conv2_2 geometry: 128 input channels, eight waves of 16 output channels
    (0, 1, 16, 32, 128, 64, 3, 1, 1, 0, 0, False),    # 128 -> 64 (the shape of conv2_1's backward-data), four waves of 16 channels
    (0, 2, 2, 2, 512, 256, 4, 2, 1, 1, 0, True),      # bottleneck: 2 pixels, split-K
    (0, 5, 4, 4, 256, 128, 4, 2, 1, 1, 0, True),      # 20 pixels: 32-pixel tile
    (1, 2, 4, 4, 32, 16, 4, 2, 1, 2, 0, True),        # deconv
    (1, 2, 8, 8, 128, 4, 4, 2, 1, 2, 3, True),        # decoder_1-like: 4 channels + tanh
    (1, 2, 1, 1, 512, 512, 4, 2, 1, 2, 0, True),      # merged_decoder_5-like: 1x1 -> 2x2, split-K
    (1, 3, 16, 16, 64, 64, 4, 2, 1, 2, 0, False),
    (0, 2, 24, 40, 64, 128, 3, 1, 1, 0, 2, False),    # patch kernel: 3x3 s1, ragged 8x16 tiles
    (0, 3, 17, 19, 32, 64, 4, 1, 1, 0, 0, False),     # patch kernel: 4x4 s1 (D layer_4 geometry), odd sizes
    (0, 1, 32, 32, 256, 256, 3, 1, 1, 0, 2, False),   # patch kernel: many channel chunks
    (0, 1, 256, 256, 32, 256, 3, 1, 1, 0, 2, False),  # 65536 pixels x 256 channels: the 256x256 wave-specialised tile
    (0, 2, 128, 128, 64, 128, 4, 2, 1, 0, 0, False),  # encoder_2 geometry: weight gradient on the 256x128 8-wave tile
    (0, 8, 64, 64, 64, 256, 4, 2, 1, 0, 0, False),    # weight gradient on the 256x256 8-wave tile
    (0, 2, 256, 256, 32, 128, 3, 1, 1, 0, 2, False),  # 131072 pixels x 128 channels: the 128x512 register-double-buffered tile
    (0, 4, 256, 256, 64, 256, 4, 2, 1, 0, 0, False),  # 65536 pixels x 256 channels, 4x4 stride 2, K = 1024: 256x256 double-buffered tile
    # stride-1 patch kernel (conv_patch.hip; the fixture below lets it run on small grids): ragged 2-D tiles, every tile shape
    (0, 2, 40, 48, 64, 256, 3, 1, 1, 0, 2, False),    # 256 ch x 16x16 px tiles, 3x3, ragged in both directions, two channel chunks
    (0, 2, 33, 70, 32, 128, 4, 1, 1, 0, 0, False),    # 128 ch x 16x32 px tiles, 4x4 stride 1 (32 x 69 outputs), one chunk
    (0, 1, 48, 64, 128, 64, 3, 1, 1, 0, 2, False),    # 64 ch x 16x32 px tiles (half-instruction weight DMAs), four chunks
    (0, 3, 31, 31, 64, 64, 4, 1, 1, 0, 0, False),     # D layer_4 geometry at a small width: 30 x 30 outputs
    # transpose-read weight gradient (wgrad_tr.hip): 16 * Cin a multiple of 256, >= 128 output channels, plain operands
    (0, 3, 17, 19, 64, 128, 4, 1, 1, 0, 0, False),    # odd 16 x 18 output grid inside a 16 x 32 padded K grid, stride 1
    (1, 2, 8, 8, 128, 64, 4, 2, 1, 0, 0, False),      # transposed conv: the gathered operand is dY
    # thin-output tile kernels (input halo staged once in LDS): power-of-two grids of at least 4 x 16 base pixels, plain operands, and
    # large enough that the planner keeps split-K at 1 (decoder_1's f32-output kernel only exists inside the step: test_gpu_step.py)
    (0, 4, 64, 128, 8, 64, 3, 1, 1, 0, 2, False),     # conv1_1 geometry: its backward-data runs conv3x3_cout8_tile_kernel
    (0, 4, 128, 128, 8, 64, 4, 2, 1, 0, 0, False),    # layer_1 geometry: its backward-data runs deconv_cout8_tile_kernel (64 x 64 dY grid)
]


# Cases whose only purpose is to put a kernel class of the benchmark plans (tests/test_gpu_coverage.py) under the oracle: each is
# sized so the planner makes the same family / tile decision as for the named layer of the bs-32 / bs-8 / bs-4 plans
# (patch_min_blocks the case is planned with, case)
CLASS_CASES = [
    (384, (0, 8, 256, 256, 32, 128, 4, 2, 1, 0, 0, False)),   # 131072 px x 128 ch, 4x4 stride 2 (layer_2 / encoder_2 fwd at batch 32): igemm_dma 128x256
    (384, (0, 8, 96, 96, 256, 512, 4, 1, 1, 0, 0, False)),    # 72200 px x 512 ch, K = 4096 (D layer_4 forward): igemm_ws 256x256; its backward-data: patch 256x128
    (1, (0, 2, 64, 64, 128, 128, 4, 2, 1, 0, 0, False)),      # backward-data of a 128 -> 128 stride-2 conv (layer_3 geometry): patch2 128x256
    (384, (0, 2, 8, 8, 64, 128, 4, 2, 1, 0, 0, False)),       # 4 x 4 output grid (fewer than 32 K slots per image): the generic wgrad_tr 256x128
    (384, (0, 8, 128, 128, 64, 128, 4, 2, 1, 0, 0, False)),   # 512 tiles of 4 x 16 pixels (layer_2 / encoder_2 forward from 8 frames): register-resident weights, conv_s2c64.hip
    (384, (0, 2, 192, 192, 8, 128, 4, 2, 1, 0, 0, False)),     # 144 blocks of 128 x 128 (above the small-grid rule of round 6, vp_tune "igemm_small_grid"), 8-channel taps (no scalar K stepping in either dtype): igemm_dma 128x128
    (384, (1, 8, 64, 64, 256, 64, 4, 2, 1, 0, 0, False)),     # merged2_decoder_2 geometry (256 -> 64 transposed conv, 512 tiles per row parity): conv_dc256_kernel; its backward-data: the two-output s2c64 form needs the step (test_gpu_step.py)
]


@pytest.fixture(autouse=True)
def small_grids_on_the_patch_kernel():
  L = _lib.lib()
  L.vp_tune(b"patch_min_blocks", 1)
  yield
  L.vp_tune(b"patch_min_blocks", -1)       # (< 0: the library default)


def make_case(case, seed=0):
  kind, n, h, w, cin, cout, k, s, p, in_act, out_act, affine = case
  rng = np.random.default_rng(seed)
  x = rng.normal(size=(n, h, w, cin))
  wt = rng.normal(0, 0.05, (k, k, cin, cout) if kind == 0 else (4, 4, cout, cin))
  b = rng.normal(0, 0.1, cout)
  sc = rng.normal(1, 0.2, cin) if affine else None
  sh = rng.normal(0, 0.2, cin) if affine else None
  return x, wt, b, sc, sh


def ref_input(x, sc, sh, in_act, dtype):
  xr = gu.rounded(x, dtype)
  if sc is not None:
    xr = np.float32(sc) * xr + np.float32(sh)
  xa = ACTS[in_act](xr)
  return gu.rounded(xa, dtype) if dtype == "bf16" else xa


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", FWD_CASES)
def test_conv_fwd(case, dtype):
  kind, n, h, w, cin, cout, k, s, p, in_act, out_act, affine = case
  x, wt, b, sc, sh = make_case(case)
  d = gu.conv_desc(kind, n, h, w, cin, cout, k, s, p, dtype, in_act, out_act)
  y = gu.conv_fwd(d, x, sc, sh, wt, b, dtype)
  xa = ref_input(x, sc, sh, in_act, dtype)
  wr = gu.rounded(wt, dtype)
  yr = ops.conv2d_fwd(xa, wr, np.float32(b).astype(np.float64), s, p) if kind == 0 else ops.deconv4s2_fwd(xa, wr, np.float32(b).astype(np.float64))
  yr = ACTS[out_act](yr)
  assert np.isfinite(y).all()
  assert gu.rel_l2(y, yr) < TOL[dtype], gu.rel_l2(y, yr)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("small", [0, 1])
@pytest.mark.parametrize("case", [c for c in FWD_CASES if c[6] in (3, 4) and c[7] == 1 and c[4] >= 32 and c[5] >= 64 and not c[11] and c[9] == 0])
def test_patch_kernel_tile_variants(case, small, dtype):
  """The other tile shapes of the stride-1 patch kernel (test_conv_fwd runs the default): small = 0: one 8-wave block per CU
  (16x16 / 16x32 pixel tiles), small = 1: two blocks per CU for the 128- / 64-row tiles only (the default, 3, adds 8x16 pixels for
  the 256-row tile)."""
  L = _lib.lib()
  L.vp_tune(b"patch_small_tiles", small)
  try:
    test_conv_fwd(case, dtype)
  finally:
    L.vp_tune(b"patch_small_tiles", 3)


BWD_CASES = [c for c in FWD_CASES if c[5] >= 8 and (c[5] & (c[5] - 1)) == 0]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", BWD_CASES)
def test_conv_bwd_data(case, dtype):
  kind, n, h, w, cin, cout, k, s, p, in_act, out_act, affine = case
  x, wt, b, sc, sh = make_case(case)
  d = gu.conv_desc(kind, n, h, w, cin, cout, k, s, p, dtype)
  ho, wo = gu.out_hw(d)
  dy = np.random.default_rng(7).normal(size=(n, ho, wo, cout))
  dx = gu.conv_bwd_data(d, dy, wt, dtype)
  dyr, wr = gu.rounded(dy, dtype), gu.rounded(wt, dtype)
  if kind == 0:
    dxr, _, _ = ops.conv2d_bwd(np.zeros((n, h, w, cin)), wr, dyr, s, p, need_dw=False)
  else:
    dxr, _, _ = ops.deconv4s2_bwd(np.zeros((n, h, w, cin)), wr, dyr)
  assert gu.rel_l2(dx, dxr) < TOL[dtype], gu.rel_l2(dx, dxr)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", BWD_CASES)
def test_conv_bwd_weight(case, dtype):
  kind, n, h, w, cin, cout, k, s, p, in_act, out_act, affine = case
  x, wt, b, sc, sh = make_case(case)
  d = gu.conv_desc(kind, n, h, w, cin, cout, k, s, p, dtype, in_act)
  ho, wo = gu.out_hw(d)
  dy = np.random.default_rng(7).normal(size=(n, ho, wo, cout))
  dw = gu.conv_bwd_weight(d, x, sc, sh, dy, wt.shape, dtype)
  xa = ref_input(x, sc, sh, in_act, dtype)
  dyr = gu.rounded(dy, dtype)
  if kind == 0:
    _, dwr, _ = ops.conv2d_bwd(xa, np.zeros_like(wt), dyr, s, p, need_dx=False)
  else:
    _, dwr, _ = ops.deconv4s2_bwd(xa, np.zeros_like(wt), dyr, need_dx=False)
  assert gu.rel_l2(dw, dwr) < TOL[dtype], gu.rel_l2(dw, dwr)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("pixels,c", [(1, 64), (37, 8), (4096, 128), (3001, 512)])
def test_batchnorm_stats_and_backward(pixels, c, dtype):
  L = _lib.lib()
  rng = np.random.default_rng(3)
  y = rng.normal(0.3, 1.5, (pixels, c))
  dz = rng.normal(size=(pixels, c))
  gamma, beta = rng.normal(1, 0.1, c), rng.normal(0, 0.1, c)
  yd, dzd = gu.to_dev(y, dtype), gu.to_dev(dz, dtype)
  g, b = gu.dev_f32(gamma), gu.dev_f32(beta)
  outs = [torch.zeros(c, device="cuda") for _ in range(6)]
  ws = torch.zeros(L.vp_bn_workspace_bytes(pixels, c, gu.VP_BF16 if dtype == "bf16" else gu.VP_F32), dtype=torch.uint8, device="cuda")
  dt = gu.VP_BF16 if dtype == "bf16" else gu.VP_F32
  _lib.check(L.vp_bn_stats(gu.ptr(yd), pixels, c, dt, gu.ptr(g), gu.ptr(b), 1e-5, gu.ptr(outs[0]), gu.ptr(outs[1]),
                           gu.ptr(outs[2]), gu.ptr(outs[3]), gu.ptr(ws), gu.stream()))
  dyd = torch.empty_like(dzd)
  _lib.check(L.vp_bn_bwd(gu.ptr(yd), gu.ptr(dzd), gu.ptr(dyd), pixels, c, dt, gu.ptr(g), gu.ptr(outs[2]), gu.ptr(outs[3]),
                         gu.ptr(outs[4]), gu.ptr(outs[5]), gu.ptr(ws), gu.stream()))
  torch.cuda.synchronize()
  yr, dzr = gu.rounded(y, dtype).reshape(1, 1, pixels, c), gu.rounded(dz, dtype).reshape(1, 1, pixels, c)
  g32, b32 = np.float32(gamma).astype(np.float64), np.float32(beta).astype(np.float64)
  z, cache = ops.bn_train_fwd(yr, g32, b32)
  scale, shift = outs[0].cpu().numpy(), outs[1].cpu().numpy()
  zd = scale * yr + shift
  if pixels == 1:   # zero variance: z == beta exactly (SURVEY 3.3 edge case)
    np.testing.assert_array_equal(zd.reshape(-1), np.float32(beta))
    return
  assert gu.rel_l2(zd, z) < 1e-5
  dyr, dgr, dbr = ops.bn_train_bwd(dzr, cache)
  assert gu.rel_l2(dyd.float().cpu().numpy(), dyr.reshape(pixels, c)) < TOL[dtype]
  assert gu.rel_l2(outs[4].cpu().numpy(), dgr) < 1e-4
  assert gu.rel_l2(outs[5].cpu().numpy(), dbr) < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("minblk,case", CLASS_CASES)
def test_more_kernel_classes(minblk, case, dtype):
  """Forward, backward-data and backward-weight parity of the class cases (see CLASS_CASES)."""
  L = _lib.lib()
  L.vp_tune(b"patch_min_blocks", minblk)
  test_conv_fwd(case, dtype)
  if case[5] >= 8 and (case[5] & (case[5] - 1)) == 0:
    test_conv_bwd_data(case, dtype)
    test_conv_bwd_weight(case, dtype)


def _profile_classes(fn):
  import json
  L = _lib.lib()
  L.vp_profile_enable(1)
  try:
    fn()
    torch.cuda.synchronize()
    n = L.vp_profile_collect(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    L.vp_profile_collect(buf, len(buf))
  finally:
    L.vp_profile_enable(0)
  return {r["name"] for r in json.loads(buf.value.decode())}


@pytest.mark.parametrize("case", [(0, 2, 128, 128, 64, 128, 4, 2, 1, 0, 0, False),     # one tile per block, half the CUs
                                  (0, 3, 32, 64, 64, 128, 4, 2, 1, 0, 0, False),       # 16 x 32 output grid: 24 tiles; image borders in most patches
                                  (0, 5, 64, 160, 64, 128, 4, 2, 1, 0, 0, False)])     # 400 tiles on 256 blocks: a second trip for some, grid not a multiple of 8 x tiles
def test_stride2_conv_with_register_resident_weights(case):
  """conv_s2c64.hip below its default size threshold (vp_tune("s2c64", tiles): the smallest launch it takes): parity with the oracle as
  test_conv_fwd, and the profile shows it is what ran."""
  L = _lib.lib()
  L.vp_tune(b"s2c64", 1)
  try:
    classes = _profile_classes(lambda: test_conv_fwd(case, "bf16"))
  finally:
    L.vp_tune(b"s2c64", 512)
  assert any(c.startswith("s2c64_") for c in classes), classes


def test_weight_gradients_under_the_k_split_of_rounds_2_to_5():
  """wgrad_tr.hip's 256 x 256 tile needs (256 x 256 tiles) x (K splits) >= 256 blocks.  Since round 6 the K split prices a slab by its
  tiles (vp_tune "wgrad_slab_tile_x1000", profiles/r06_ab_wgrad_split_cost.txt): op-sized cases no longer split 32-64 ways, the batch-32 /
  512 x 512 plans still reach the tile with their pixel counts.  Every weight-gradient case again under the split of rounds 2-5 (parity
  with the oracle as test_conv_bwd_weight), and the profile shows that the 256 x 256 tile with the power-of-two K grid is among what ran
  (tests/test_gpu_coverage.py counts these classes as covered on the strength of this test)."""
  L = _lib.lib()
  cases = [c for c in FWD_CASES + [c for _, c in CLASS_CASES] if c[5] >= 8 and (c[5] & (c[5] - 1)) == 0]
  L.vp_tune(b"wgrad_slab_tile_x1000", 0)
  try:
    classes = _profile_classes(lambda: [test_conv_bwd_weight(c, "bf16") for c in cases])
  finally:
    L.vp_tune(b"wgrad_slab_tile_x1000", -1)
  assert "wgrad_tr_exact_bf16_256x256" in classes, sorted(classes)


def test_thin_layer_cases_run_on_their_dedicated_kernels():
  """The parity cases above are only worth something if the dedicated kernels are what runs: check the profile's class names."""
  l1 = (0, 4, 128, 128, 8, 64, 4, 2, 1, 0, 0, False)
  assert any(c.startswith("dcout8_") for c in _profile_classes(lambda: test_conv_bwd_data(l1, "bf16")))
  c11 = (0, 4, 64, 128, 8, 64, 3, 1, 1, 0, 2, False)
  assert any(c.startswith("cout8_") for c in _profile_classes(lambda: test_conv_bwd_data(c11, "bf16")))
  assert any(c.startswith("cin8_") for c in _profile_classes(lambda: test_conv_fwd(c11, "bf16")))
