"""Oracle: one training sample of PixReferDataGenerator.iterator (reference generator/generator.py:975-1019), restated on the host.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the fixture generator (tests/golden/make_golden.py `frames`) and
tests/test_gpu_input_pipeline.py check the device kernel `vp_pixrefer_pack_frames` against this file; nothing under voicepuppet_amd/
imports it.  Written from the reference's lines, independently of the product's own host pipeline
(voicepuppet_amd/generator/generator.py) and of csrc/pointwise.hip's frame_pack_kernel.

UNPINNED BY cv2: `cv2.resize` (generator.py:989,1001: INTER_LINEAR on float planes - the frames were divided by 255 by
ImageLoader.get_data, loader.py) is not installable here; PIL's mode-"F" BILINEAR resize stands in for it.  For the crop sizes the
generator draws (rsize <= img_size, i.e. enlargement) both are a plain two-tap bilinear interpolation with half-pixel centres and edge
clamping; the device kernel is compared within 2e-6 absolute.

Per sample (generator.py lines in brackets):
  frame  = imread(jpg) / 255, BGR -> RGB                                              [983-984, 996-997]
  frame  = concat(target | 3dface | mask, axis = channel)  -> [S, S, 9]               [985-988, 998]
  frame  = frame[rx : rx + rsize, ry : ry + rsize]  (rx indexes ROWS, as written)     [989, 999]
  frame  = resize(frame, (S, S))                                                      [990, 1000]
  frame  = concat back along the width -> [S, 3S, 3]                                  [991-994, 1001]
  inputs    = the two 3-D-face thirds, (example, current) interleaved on channels -> [S, S, 6]     [1008-1010]
  fg_inputs = (targets * masks) of (example, current), the same interleave                         [1011-1014]
  targets, masks = the current frame's thirds                                                      [1016]
"""
import numpy as np


def _triptych(u8_bgr, crop, S):
  from PIL import Image
  rx, ry, rsize = (int(v) for v in crop)
  rgb = u8_bgr[:, :, ::-1].astype(np.float32) / np.float32(255.0)
  nine = np.concatenate([rgb[:, 0:S], rgb[:, S:2 * S], rgb[:, 2 * S:3 * S]], axis=2)
  nine = nine[rx:rx + rsize, ry:ry + rsize]
  planes = []
  for c in range(9):
    im = Image.fromarray(np.ascontiguousarray(nine[:, :, c]), mode="F")
    planes.append(np.asarray(im.resize((S, S), Image.BILINEAR), dtype=np.float32))
  nine = np.stack(planes, axis=2)
  return np.concatenate([nine[:, :, 0:3], nine[:, :, 3:6], nine[:, :, 6:9]], axis=1)


def pack_frames_ref(ex_u8, cur_u8, crops, img_size):
  """ex_u8 / cur_u8: [S, 3S, 3] uint8 BGR triptych frames (example, current); crops: [2, 3] (rx, ry, rsize) for each.
  Returns (inputs [S,S,6], fg_inputs [S,S,6], targets [S,S,3], masks [S,S,3]) float32."""
  S = img_size
  pair = np.stack([_triptych(ex_u8, crops[0], S), _triptych(cur_u8, crops[1], S)])      # [2, S, 3S, 3]
  tgt, face, msk = pair[:, :, 0:S], pair[:, :, S:2 * S], pair[:, :, 2 * S:3 * S]
  inputs = np.concatenate([face[0], face[1]], axis=2)
  fg = tgt * msk
  fg_inputs = np.concatenate([fg[0], fg[1]], axis=2)
  return inputs, fg_inputs, tgt[1], msk[1]
