"""Drop-in for the reference's utils/reconstruct_mesh.py on the MI355X (libvp_hip.so: vp_bfm_reconstruct): BFM coefficients ->
face shape / texture / colour / projection, batched over the frames of a clip, plus `ClipRenderer`, which chains it with the
rasteriser (utils.mesh_core) the way render_face does per frame (voicepuppet/pixrefer/infer_bfmvid.py:79-108).

Host work kept here on purpose: the one-time promotion of the face model to float64 device arrays, the mean-shape centre and
SH constants (model load), and the 3x3 rotation matrices of the clip (numpy, same libm as the reference).  No CPU fallback.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from . import mesh_core


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def Split_coeff(coeff):
  """reconstruct_mesh.py:5-13."""
  return coeff[:, :80], coeff[:, 80:144], coeff[:, 144:224], coeff[:, 224:227], coeff[:, 227:254], coeff[:, 254:]


def Compute_rotation_matrix(angles):
  """reconstruct_mesh.py:68-93 for [T,3] angles (the reference takes T=1): float64 [T,3,3] = (Rz Ry Rx)^T, all frames at once."""
  angles = np.asarray(angles)
  T = angles.shape[0]
  c, s = np.cos(angles), np.sin(angles)                      # same dtype promotion as the reference's scalar calls
  one, zero = np.ones(T), np.zeros(T)
  rx = np.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], 1).reshape(T, 3, 3)
  ry = np.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], 1).reshape(T, 3, 3)
  rz = np.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], 1).reshape(T, 3, 3)
  return np.ascontiguousarray(np.transpose(np.matmul(np.matmul(rz, ry), rx), (0, 2, 1)))


class DeviceFaceModel:
  """The reference's `BFM` object (utils/bfm_load_data.py:9-21; any object with those attributes) resident in HBM."""

  def __init__(self, facemodel, device=None, focal=1015.0, center=112.0):
    if not torch.cuda.is_available():
      raise RuntimeError("BFM reconstruction needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    f64 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float64))).to(dev)
    self.device = dev
    self.meanshape = f64(np.asarray(facemodel.meanshape).reshape(-1))
    self.nver = self.meanshape.numel() // 3
    for k, width in (("idBase", 80), ("exBase", 64), ("texBase", 80)):
      if tuple(np.shape(getattr(facemodel, k))) != (3 * self.nver, width):
        raise ValueError("face model shapes must be idBase [3N,80], exBase [3N,64], texBase [3N,80], meanshape/meantex [1,3N]")
    # k-major copies ([K,3N]): every basis load of the device kernel is then a coalesced wave request
    self.idBase, self.exBase, self.texBase = (f64(np.asarray(getattr(facemodel, k)).T) for k in ("idBase", "exBase", "texBase"))
    self.meantex = f64(np.asarray(facemodel.meantex).reshape(-1))
    if self.meantex.numel() != 3 * self.nver:
      raise ValueError("meantex must be [1,3N]")
    tri = np.ascontiguousarray((np.asarray(facemodel.tri) - 1).astype(np.int32))
    pb = np.ascontiguousarray((np.asarray(facemodel.point_buf) - 1).astype(np.int32))
    self.ntri = tri.shape[0]
    if pb.shape != (self.nver, 8) or pb.min() < 0 or pb.max() > self.ntri or tri.min() < 0 or tri.max() >= self.nver:
      raise ValueError("tri / point_buf out of range (1-based indices expected, point_buf padded with ntri+1)")
    self.tri, self.point_buf = torch.from_numpy(tri).to(dev), torch.from_numpy(pb).to(dev)
    self.keypoints = np.asarray(facemodel.keypoints).astype(np.int64)
    m = _lib.BfmModel()
    m.nver, m.ntri = self.nver, self.ntri
    for k in ("meanshape", "idBase", "exBase", "meantex", "texBase", "tri", "point_buf"):
      setattr(m, k, getattr(self, k).data_ptr())
    cen = np.mean(np.reshape(np.asarray(facemodel.meanshape), [1, -1, 3]), axis=1, keepdims=True).reshape(3)   # reconstruct_mesh.py:27
    a0, a1, a2 = np.pi, 2 * np.pi / np.sqrt(3.0), 2 * np.pi / np.sqrt(8.0)
    c0, c1, c2 = 1 / np.sqrt(4 * np.pi), np.sqrt(3.0) / np.sqrt(4 * np.pi), 3 * np.sqrt(5.0) / np.sqrt(12 * np.pi)
    sh = [a0 * c0, a1 * c1, a2 * c2, a2 * c2 * 0.5 / np.sqrt(3.0), a2 * c2 * 0.5]
    for i in range(3):
      m.center[i] = float(cen[i])
    for i in range(5):
      m.sh[i] = float(sh[i])
    m.focal, m.image_center = float(focal), float(center)
    self.c = m
    self._ws = None

  def workspace(self, frames):
    n = _lib.lib().vp_bfm_reconstruct_workspace_bytes(self.nver, self.ntri, frames)
    if self._ws is None or self._ws.numel() < n:
      self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
    return self._ws


def reconstruct_clip(coeff, model, angles, shared_texture=False, full=True):
  """Batched Reconstruction_rotation.  coeff [T,257] (numpy or CUDA float32), angles [T,3]; model a DeviceFaceModel.
  Returns a dict of CUDA tensors: vertices / colors (float32, the rasteriser's inputs) and, with full=True, the float64
  face_shape, face_texture, face_color, face_projection, z_buffer, landmarks_2d of the reference."""
  dev = model.device
  coeff_d = (torch.from_numpy(np.ascontiguousarray(coeff, np.float32)) if isinstance(coeff, np.ndarray) else coeff).to(dev).contiguous()
  T = coeff_d.shape[0]
  if coeff_d.dtype != torch.float32 or coeff_d.shape[1] != 257:
    raise ValueError("coeff must be float32 [T,257]")
  rot = torch.from_numpy(Compute_rotation_matrix(np.asarray(angles).reshape(T, 3))).to(dev)
  N = model.nver
  out = {"vertices": torch.empty(T, N, 3, dtype=torch.float32, device=dev), "colors": torch.empty(T, N, 3, dtype=torch.float32, device=dev)}
  if full:
    for k, shp in (("face_shape", (T, N, 3)), ("face_texture", (1 if shared_texture else T, N, 3)), ("face_color", (T, N, 3)),
                   ("face_projection", (T, N, 2)), ("z_buffer", (T, N, 1))):
      out[k] = torch.empty(*shp, dtype=torch.float64, device=dev)
  ws = model.workspace(T)
  L = _lib.lib()
  _lib.check(L.vp_bfm_reconstruct(ctypes.byref(model.c), _ptr(coeff_d), _ptr(rot), T, 1 if shared_texture else 0, _ptr(out.get("face_shape")),
                                  _ptr(out.get("face_texture")), _ptr(out.get("face_color")), _ptr(out.get("face_projection")),
                                  _ptr(out.get("z_buffer")), _ptr(out["vertices"]), _ptr(out["colors"]), _ptr(ws), ws.numel(), _stream()),
             "vp_bfm_reconstruct")
  if full:
    out["landmarks_2d"] = out["face_projection"][:, torch.from_numpy(model.keypoints).to(dev)]
  return out


def Reconstruction_rotation(coeff, facemodel, angles):
  """reconstruct_mesh.py:198-223, same arguments and return tuple (numpy float64, batch dimension = coeff.shape[0]).
  `facemodel` may be the reference's BFM object (uploaded on every call) or a DeviceFaceModel (resident)."""
  model = facemodel if isinstance(facemodel, DeviceFaceModel) else DeviceFaceModel(facemodel)
  o = reconstruct_clip(coeff, model, angles)
  return tuple(o[k].cpu().numpy() for k in ("face_shape", "face_texture", "face_color", "face_projection", "z_buffer", "landmarks_2d"))


class ClipRenderer:
  """render_face's reconstruction + rasterisation (infer_bfmvid.py:79-108) for all frames of a clip in four launches + one:
  images uint8 [T,h,w,3] in the reference's channel order BEFORE its cvtColor (:110), masks uint8 [T,h,w]."""

  def __init__(self, facemodel, h=224, w=224):
    self.model = facemodel if isinstance(facemodel, DeviceFaceModel) else DeviceFaceModel(facemodel)
    self.h, self.w = h, w

  def __call__(self, coeff, angles, shared_texture=True):
    o = reconstruct_clip(coeff, self.model, angles, shared_texture=shared_texture, full=False)
    T, dev = o["vertices"].shape[0], self.model.device
    image = torch.zeros(T, self.h, self.w, 3, dtype=torch.uint8, device=dev)
    mask = torch.zeros(T, self.h, self.w, dtype=torch.uint8, device=dev)
    depth = torch.full((T, self.h, self.w), -99999.0, dtype=torch.float32, device=dev)       # infer_bfmvid.py:104
    mesh_core.render_colors(image, mask, o["vertices"], self.model.tri, o["colors"], depth)
    return image, mask
