"""cv2.resize(uint8 image, dsize) [INTER_LINEAR] + paste on the device (csrc/resize.hip): the last step of the reference's
render_face (voicepuppet/pixrefer/infer_bfmvid.py:110-121), byte for byte OpenCV's fixed-point bilinear."""
import ctypes

import torch

from .. import _lib


def resize_paste_u8(src, dst_h, dst_w, canvas_shape, y0, x0, swap_rb=False, canvas=None):
  """src: uint8 device tensor [T, h, w, 3]; returns the uint8 canvas [T, H, W, 3] (zeros outside the pasted dst_h x dst_w image,
  as `back_new_image = np.zeros(...)` in the reference).  swap_rb: cv2.cvtColor(BGR2RGB) in front of the resize."""
  if not (src.is_cuda and src.dtype == torch.uint8 and src.dim() == 4 and src.shape[-1] == 3):
    raise ValueError("resize_paste_u8: src must be a uint8 device tensor [T, h, w, 3]")
  L = _lib.lib()
  src = src.contiguous()
  T, h, w, _ = src.shape
  H, W = int(canvas_shape[0]), int(canvas_shape[1])
  if canvas is None:
    canvas = torch.zeros((T, H, W, 3), dtype=torch.uint8, device=src.device)
  ws = torch.empty(int(L.vp_resize_paste_workspace_bytes(int(dst_h), int(dst_w))), dtype=torch.uint8, device=src.device)
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  _lib.check(L.vp_resize_paste_u8(ctypes.c_void_p(src.data_ptr()), T, h, w, int(dst_h), int(dst_w), 1 if swap_rb else 0,
                                  ctypes.c_void_p(canvas.data_ptr()), H, W, int(y0), int(x0), ctypes.c_void_p(ws.data_ptr()), st),
             "vp_resize_paste_u8")
  return canvas
