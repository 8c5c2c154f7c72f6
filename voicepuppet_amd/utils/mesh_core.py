"""Drop-in for the reference's `mesh_core_cython` module (utils/cython/mesh_core_cython.pyx:64-82): the flat-shaded
z-buffer rasteriser that turns BFMNet's reconstructed mesh into PixReferNet's conditioning image
(voicepuppet/pixrefer/infer_bfmvid.py:100-108), on the MI355X (libvp_hip.so: vp_render_colors).

`render_colors_core` keeps the reference's positional signature and in-place convention.  numpy arguments are staged
through HBM (the reference's calling convention); CUDA tensors are rasterised where they lie.  `render_colors` is the
batched form for a clip: F frames sharing one triangle list in ONE launch.  No CPU fallback.
"""
import ctypes

import numpy as np
import torch

from .. import _lib


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr())


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_workspaces = {}


def _workspace(nbytes, device):
  key = (device.index if device.index is not None else torch.cuda.current_device())
  ws = _workspaces.get(key)
  if ws is None or ws.numel() < nbytes:
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _workspaces[key] = ws
  return ws


def render_colors(image, face_mask, vertices, triangles, colors, depth_buffer):
  """Batched, device-resident form.  image u8 [F,h,w,c], face_mask u8 [F,h,w], vertices f32 [F,nver,3], triangles i32 [ntri,3],
  colors f32 [F,nver,c], depth_buffer f32 [F,h,w]; image / face_mask / depth_buffer are updated in place."""
  if not torch.cuda.is_available():
    raise RuntimeError("render_colors needs an MI355X (no CPU fallback)")
  F, h, w, c = image.shape
  nver = vertices.shape[1]
  ntri = triangles.shape[0]
  for t, dt in ((image, torch.uint8), (face_mask, torch.uint8), (vertices, torch.float32), (triangles, torch.int32),
                (colors, torch.float32), (depth_buffer, torch.float32)):
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dt):
      raise ValueError("render_colors: arguments must be contiguous CUDA tensors of the reference's dtypes")
  if tuple(face_mask.shape) != (F, h, w) or tuple(depth_buffer.shape) != (F, h, w) or tuple(vertices.shape) != (F, nver, 3) \
      or tuple(colors.shape) != (F, nver, c) or triangles.shape[1] != 3:
    raise ValueError("render_colors: inconsistent shapes")
  L = _lib.lib()
  nbytes = L.vp_render_colors_workspace_bytes(F, h, w)
  ws = _workspace(nbytes, image.device)
  _lib.check(L.vp_render_colors(_ptr(image), _ptr(face_mask), _ptr(vertices), _ptr(triangles), _ptr(colors), _ptr(depth_buffer),
                                ntri, nver, h, w, c, F, _ptr(ws), _stream()), "vp_render_colors")
  return image, face_mask, depth_buffer


def render_colors_core(image, face_mask, vertices, triangles, colors, depth_buffer, ntri, h, w, c):
  """mesh_core_cython.render_colors_core(image, face_mask, vertices, triangles, colors, depth_buffer, ntri, h, w, c):
  flat uint8 image [h*w*c] and face_mask [h*w], flat float32 vertices [nver*3] / colors [nver*c] / depth_buffer [h*w],
  flat int32 triangles [ntri*3]; image, face_mask and depth_buffer are overwritten in place, nothing is returned."""
  host = isinstance(image, np.ndarray)
  if host:
    for a, dt in ((image, np.uint8), (face_mask, np.uint8), (vertices, np.float32), (triangles, np.int32), (colors, np.float32),
                  (depth_buffer, np.float32)):
      if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous):
        raise ValueError("render_colors_core: C-contiguous numpy arrays of the reference's dtypes expected")
    dev = torch.device("cuda", torch.cuda.current_device())
    t = [torch.from_numpy(a).to(dev) for a in (image, face_mask, vertices, triangles, colors, depth_buffer)]
  else:
    t = [image, face_mask, vertices, triangles, colors, depth_buffer]
  nver = t[2].numel() // 3
  if t[3].numel() < 3 * ntri or t[0].numel() != h * w * c or t[1].numel() != h * w or t[5].numel() != h * w or t[4].numel() != nver * c:
    raise ValueError("render_colors_core: buffer sizes do not match ntri / h / w / c")
  render_colors(t[0].view(1, h, w, c), t[1].view(1, h, w), t[2].view(1, nver, 3), t[3].view(-1)[:3 * ntri].view(ntri, 3),
                t[4].view(1, nver, c), t[5].view(1, h, w))
  if host:
    image[...] = t[0].cpu().numpy().reshape(image.shape)
    face_mask[...] = t[1].cpu().numpy().reshape(face_mask.shape)
    depth_buffer[...] = t[5].cpu().numpy().reshape(depth_buffer.shape)
