"""Oracle: cv2.resize(uint8, dsize) with the default INTER_LINEAR, restated in numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  UNPINNED BY cv2: OpenCV is not installed in the build container and is a
dependency the reference neither vendors nor pins (voicepuppet/pixrefer/infer_bfmvid.py:111 calls cv2.resize with two positional
arguments, i.e. INTER_LINEAR on a uint8 image).  What is restated is OpenCV's published algorithm for 8-bit images
(modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeLinear<uchar,int,short,2048> and the uchar specialisation of
VResizeLinear), unchanged across 3.x / 4.x:

  scale = 1 / (dsize / ssize) in double;  fx = float32((dx + 0.5) * scale - 0.5);  sx = floor(fx);  fx -= sx   [float32]
  columns: sx < 0 -> (0, fx = 0);  sx >= width - 1 -> (width - 1, fx = 0), and such columns read ONE tap with weight 2048
  coefficients: short(rint(float32(1 - fx) * 2048)), short(rint(fx * 2048))                       [round half to even]
  rows: the floor is kept and the two source rows are clipped to [0, height - 1] instead
  pass 1 (int32): D = S[sx] * a0 + S[sx + 1] * a1
  pass 2: dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
  shortcuts: equal sizes -> copy; an exact 2x reduction in both directions -> INTER_AREA: (a + b + c + d + 2) >> 2

Pinned by hand-derived known answers (tests/test_cv_resize.py) - written independently of csrc/resize.hip (vectorised numpy in
float32 / int64 here, scalar C there).
"""
import numpy as np


def _coeffs(ssize, dsize):
  scale = 1.0 / (float(dsize) / float(ssize))
  d = np.arange(dsize, dtype=np.float64)
  f = ((d + 0.5) * scale - 0.5).astype(np.float32)
  s = np.floor(f).astype(np.int64)
  f = (f - s.astype(np.float32)).astype(np.float32)
  return s, f


def _fix(c):
  return np.clip(np.rint(c.astype(np.float32) * np.float32(2048.0)), -32768, 32767).astype(np.int64)


def resize_linear_u8(img, dsize):
  """img [h, w, c] uint8, dsize = (width, height) as cv2.resize takes it -> [height, width, c] uint8."""
  img = np.asarray(img)
  assert img.dtype == np.uint8 and img.ndim == 3
  h, w, _ = img.shape
  dw, dh = int(dsize[0]), int(dsize[1])
  if (dw, dh) == (w, h):
    return img.copy()
  if w == 2 * dw and h == 2 * dh:
    s = img.astype(np.int64)
    return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
  sx, fx = _coeffs(w, dw)
  lo, hi = sx < 0, sx >= w - 1
  fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
  sx = np.clip(sx, 0, w - 1)
  a0, a1 = _fix(np.float32(1) - fx), _fix(fx)
  a0 = np.where(hi, 2048, a0)
  a1 = np.where(hi, 0, a1)
  sx1 = np.minimum(sx + 1, w - 1)
  s = img.astype(np.int64)
  D = s[:, sx] * a0[None, :, None] + s[:, sx1] * a1[None, :, None]           # [h, dw, c]
  sy, fy = _coeffs(h, dh)
  b0, b1 = _fix(np.float32(1) - fy), _fix(fy)
  r0, r1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
  out = (((b0[:, None, None] * (D[r0] >> 4)) >> 16) + ((b1[:, None, None] * (D[r1] >> 4)) >> 16) + 2) >> 2
  return (out & 0xff).astype(np.uint8)


def render_face_tail(image_bgr, ratio, canvas_shape, center_x, center_y, tx, ty):
  """infer_bfmvid.py:110-121 on one rasterised frame: BGR2RGB, resize to round(224 / ratio), paste into a zero canvas."""
  rgb = np.ascontiguousarray(image_bgr[..., ::-1])
  side = (int(round(rgb.shape[0] / ratio)), int(round(rgb.shape[1] / ratio)))
  new = resize_linear_u8(rgb, side)
  back = np.zeros(canvas_shape, np.uint8)
  cfx, cfy = new.shape[1] // 2, new.shape[0] // 2
  ry = center_y - cfy + new.shape[0] - ty
  rx = center_x - cfx + new.shape[1] - tx
  back[center_y - cfy - ty:ry, center_x - cfx - tx:rx, :] = new
  return back
