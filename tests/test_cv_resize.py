"""cv2.resize(uint8, INTER_LINEAR) restated (oracle/cv_resize_ref.py) - UNPINNED BY cv2 (not installable here; see the oracle's
header): pinned by known answers derived by hand from OpenCV's fixed-point algorithm, by its structural properties, and (CPU) by the
agreement of the product's host-side coefficient tables (csrc/resize.hip, scalar C float arithmetic) with the oracle's (vectorised
numpy float32).  The -m gpu test compares the device kernel with the oracle byte for byte."""
import ctypes

import numpy as np
import pytest

from oracle import cv_resize_ref as cr


def test_known_answers_derived_by_hand():
  # [0, 100] -> 4 columns: scale 0.5; fx = -0.25 (clamped: S[0]), 0.25, 0.75, 1.25 (last column: S[1] alone)
  #   pass 1: 0, 0*1536 + 100*512 = 51200, 0*512 + 100*1536 = 153600, 100*2048 = 204800
  #   pass 2 (b0 = 2048): ((2048 * (D >> 4)) >> 16) = 0, 100, 300, 400 -> (+2) >> 2 = 0, 25, 75, 100
  a = np.array([[[0], [100]]], np.uint8)
  np.testing.assert_array_equal(cr.resize_linear_u8(a, (4, 1))[0, :, 0], [0, 25, 75, 100])
  # [10..60] (6 columns) -> 4: scale 1.5; (sx, fx) = (0, .25), (1, .75), (3, .25), (4, .75)
  #   10*1536 + 20*512 = 25600 -> 50 -> 13;  20*512 + 30*1536 = 56320 -> 110 -> 28;  87040 -> 170 -> 43;  117760 -> 230 -> 58
  b = np.arange(10, 70, 10, dtype=np.uint8).reshape(1, 6, 1)
  np.testing.assert_array_equal(cr.resize_linear_u8(b, (4, 1))[0, :, 0], [13, 28, 43, 58])
  # the same numbers along rows: the row pass clips source rows instead of clamping the coefficient, same result here
  np.testing.assert_array_equal(cr.resize_linear_u8(a.transpose(1, 0, 2), (1, 4))[:, 0, 0], [0, 25, 75, 100])
  np.testing.assert_array_equal(cr.resize_linear_u8(b.transpose(1, 0, 2), (1, 4))[:, 0, 0], [13, 28, 43, 58])
  # 2-D: separable -> outer structure; 255 stays 255 (no overflow in the 16-bit products), 0 stays 0
  c = np.full((3, 5, 3), 255, np.uint8)
  assert (cr.resize_linear_u8(c, (7, 11)) == 255).all()
  assert (cr.resize_linear_u8(np.zeros((3, 5, 3), np.uint8), (7, 11)) == 0).all()


def test_shortcuts_and_structure():
  rng = np.random.default_rng(0)
  img = rng.integers(0, 256, (8, 12, 3)).astype(np.uint8)
  np.testing.assert_array_equal(cr.resize_linear_u8(img, (12, 8)), img)                      # equal sizes: copy
  half = cr.resize_linear_u8(img, (6, 4))                                                    # exact 2x reduction: INTER_AREA
  s = img.astype(np.int64)
  np.testing.assert_array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
  for dsize in ((20, 5), (7, 13), (36, 24), (11, 8)):
    out = cr.resize_linear_u8(img, dsize)
    assert out.shape == (dsize[1], dsize[0], 3) and out.dtype == np.uint8
    # every output lies between the min and max of the image (convex weights, rounding included)
    assert out.min() >= img.min() and out.max() <= img.max()
    # channels are independent
    np.testing.assert_array_equal(out[..., 1], cr.resize_linear_u8(img[..., 1:2], dsize)[..., 0])
    # within one grey level of a float bilinear interpolation at OpenCV's sample positions
    fy = np.clip((np.arange(dsize[1]) + 0.5) * img.shape[0] / dsize[1] - 0.5, 0, img.shape[0] - 1)
    fx = np.clip((np.arange(dsize[0]) + 0.5) * img.shape[1] / dsize[0] - 0.5, 0, img.shape[1] - 1)
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    y1, x1 = np.minimum(y0 + 1, img.shape[0] - 1), np.minimum(x0 + 1, img.shape[1] - 1)
    wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - wx) + f[y0][:, x1] * wx) * (1 - wy) + (f[y1][:, x0] * (1 - wx) + f[y1][:, x1] * wx) * wy
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0


def test_render_face_tail_geometry():
  """infer_bfmvid.py:110-121: channel swap, side = round(224 / ratio), paste centred at (center_x, center_y) shifted by (tx, ty)."""
  rng = np.random.default_rng(1)
  face = rng.integers(0, 256, (224, 224, 3)).astype(np.uint8)
  back = cr.render_face_tail(face, 0.9, (512, 512, 3), 250, 260, -3, 5)
  side = int(round(224 / 0.9))
  y0, x0 = 260 - side // 2 - 5, 250 - side // 2 + 3
  np.testing.assert_array_equal(back[y0:y0 + side, x0:x0 + side], cr.resize_linear_u8(face[..., ::-1].copy(), (side, side)))
  mask = np.ones((512, 512), bool)
  mask[y0:y0 + side, x0:x0 + side] = False
  assert (back[mask] == 0).all()


def test_product_coefficient_tables_equal_the_oracle():
  """vp_resize_linear_table (host code of csrc/resize.hip, no GPU call) against the oracle's coefficients for every size pair the
  render path can produce (224 -> round(224 / ratio)) and a sweep of others."""
  from voicepuppet_amd import _lib
  L = _lib.lib()
  pairs = [(224, d) for d in range(120, 420, 7)] + [(s, d) for s in (1, 2, 3, 17, 64, 223) for d in (1, 2, 5, 16, 33, 100, 447)]
  for ssize, dsize in pairs:
    ofs = (ctypes.c_int * dsize)(); a0 = (ctypes.c_short * dsize)(); a1 = (ctypes.c_short * dsize)(); r1 = (ctypes.c_int * dsize)()
    assert L.vp_resize_linear_table(ssize, dsize, 0, ofs, a0, a1, r1) == 0
    s, f = cr._coeffs(ssize, dsize)
    lo, hi = s < 0, s >= ssize - 1
    f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
    want_ofs = np.clip(s, 0, ssize - 1)
    want_a0 = np.where(hi, 2048, cr._fix(np.float32(1) - f))
    want_a1 = np.where(hi, 0, cr._fix(f))
    np.testing.assert_array_equal(np.array(ofs[:]), want_ofs)
    np.testing.assert_array_equal(np.array(a0[:]), want_a0)
    np.testing.assert_array_equal(np.array(a1[:]), want_a1)
    assert L.vp_resize_linear_table(ssize, dsize, 1, ofs, a0, a1, r1) == 0
    s, f = cr._coeffs(ssize, dsize)
    np.testing.assert_array_equal(np.array(ofs[:]), np.clip(s, 0, ssize - 1))
    np.testing.assert_array_equal(np.array(r1[:]), np.clip(s + 1, 0, ssize - 1))
    np.testing.assert_array_equal(np.array(a0[:]), cr._fix(np.float32(1) - f))
    np.testing.assert_array_equal(np.array(a1[:]), cr._fix(f))


@pytest.mark.gpu
@pytest.mark.parametrize("hs,ws,dh,dw", [(224, 224, 249, 249), (224, 224, 187, 187), (224, 224, 112, 112), (224, 224, 224, 224),
                                         (37, 53, 90, 41), (64, 48, 20, 100), (5, 7, 31, 3), (224, 224, 448, 448)])
def test_device_resize_paste_is_bit_exact(hs, ws, dh, dw):
  import torch
  from voicepuppet_amd.utils.cv_resize import resize_paste_u8
  rng = np.random.default_rng(hs * 1000 + dw)
  T = 3
  src = rng.integers(0, 256, (T, hs, ws, 3)).astype(np.uint8)
  src[0, : hs // 2] = 255                                         # saturated region
  H, W = dh + 40, dw + 30
  for swap in (False, True):
    got = resize_paste_u8(torch.tensor(src, device="cuda"), dh, dw, (H, W), 17, 9, swap_rb=swap).cpu().numpy()
    for t in range(T):
      s = src[t, ..., ::-1].copy() if swap else src[t]
      want = np.zeros((H, W, 3), np.uint8)
      want[17:17 + dh, 9:9 + dw] = cr.resize_linear_u8(s, (dw, dh))
      np.testing.assert_array_equal(got[t], want)


@pytest.mark.gpu
def test_render_faces_uses_the_exact_resize():
  """voicepuppet_amd/pixrefer/infer_bfmvid.render_faces against oracle.cv_resize_ref.render_face_tail on the frames of a fake renderer."""
  import torch
  from voicepuppet_amd.pixrefer import infer_bfmvid as ib
  rng = np.random.default_rng(3)
  frames = rng.integers(0, 256, (4, 224, 224, 3)).astype(np.uint8)

  def fake_renderer(coeff, angles):
    return torch.tensor(frames, device="cuda"), None
  tp = np.array([0, 0, 1.1, 12.0, -7.0], np.float32)
  out = ib.render_faces(fake_renderer, 256, 250, 0.8, np.zeros((4, 257), np.float32), (512, 512, 3), tp)
  ratio = 0.8 * tp[2]
  tx, ty = -int(tp[3] / ratio), -int(tp[4] / ratio)
  for i in range(4):
    np.testing.assert_array_equal(out[i], cr.render_face_tail(frames[i], ratio, (512, 512, 3), 256, 250, tx, ty))
