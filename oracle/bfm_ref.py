"""Oracle for the BFM reconstruction step that feeds the rasteriser: coefficients -> projected vertices and
per-vertex colours, batched over the frames of a clip.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates utils/reconstruct_mesh.py `Reconstruction_rotation` (:198-223)
and the packing of infer_bfmvid.py:92-99 in float64 numpy, vectorised over T frames.  PINNED: tests/golden/bfm_recon.npz
holds the outputs of the reference's own reconstruct_mesh.py (pure numpy, imported from /root/reference by
tests/golden/make_golden.py in the build container) on a synthetic face model; tests/test_bfm_recon.py checks this file
against them.
"""
import numpy as np


class FaceModel:
  """Field names of utils/bfm_load_data.py:9-21 (`BFM`): tri / point_buf are 1-based, keypoints 0-based."""

  def __init__(self, meanshape, idBase, exBase, meantex, texBase, point_buf, tri, keypoints):
    self.meanshape, self.idBase, self.exBase, self.meantex, self.texBase = meanshape, idBase, exBase, meantex, texBase
    self.point_buf, self.tri, self.keypoints = point_buf, tri, keypoints


def synthetic_facemodel(seed=0, nlat=14, nlon=18, dtype=np.float64, smooth=False):
  """A front half-ellipsoid 'face' of nlat*nlon vertices at the BFM's scale (decimetres, |x|,|y| <~ 1).  Bases: white noise
  per vertex (default; fine for the small parity fixtures) or, with smooth=True, low-frequency deformation fields like the
  real PCA bases, so that a BFM-sized mesh keeps sub-pixel triangles (the benchmark geometry)."""
  rng = np.random.default_rng(seed)
  th = np.linspace(0.25, np.pi - 0.25, nlat)
  ph = np.linspace(0.2, np.pi - 0.2, nlon)
  T, P = np.meshgrid(th, ph, indexing="ij")
  xyz = np.stack([0.75 * np.sin(T) * np.cos(P), 0.9 * np.cos(T), 0.6 * np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
  n = xyz.shape[0]
  tris = []
  for i in range(nlat - 1):
    for j in range(nlon - 1):
      a, b, c, d = i * nlon + j, i * nlon + j + 1, (i + 1) * nlon + j, (i + 1) * nlon + j + 1
      tris += [[a, c, b], [b, c, d]]
  tri = np.array(tris, np.int64) + 1
  nf = tri.shape[0]
  point_buf = np.full((n, 8), nf + 1, np.int64)                    # pad = index of the appended zero normal (reconstruct_mesh.py:47-49)
  fill = np.zeros(n, np.int64)
  for f in range(nf):
    for v in tri[f] - 1:
      point_buf[v, fill[v]] = f + 1
      fill[v] += 1
  def field(k, amp):
    if not smooth:
      return amp * rng.normal(size=(3 * n, k))
    a, c = rng.integers(1, 5, size=k), rng.integers(1, 5, size=k)
    b, d = rng.uniform(0, 2 * np.pi, size=k), rng.uniform(0, 2 * np.pi, size=k)
    wave = np.sin(T.reshape(-1, 1) * a + b) * np.cos(P.reshape(-1, 1) * c + d)          # [n,k]
    return (amp * wave[:, None, :] * rng.normal(size=(1, 3, k))).reshape(3 * n, k)
  idb, exb = field(80, 0.02), field(64, 0.03)
  meantex = rng.uniform(90, 200, size=(1, 3 * n))
  texb = field(80, 4.0)
  return FaceModel(meanshape=xyz.reshape(1, -1).astype(dtype), idBase=idb.astype(dtype), exBase=exb.astype(dtype),
                   meantex=meantex.astype(dtype), texBase=texb.astype(dtype),
                   point_buf=point_buf, tri=tri, keypoints=rng.choice(n, 68, replace=False).astype(np.int32))


def synthetic_coeffs(frames, seed=0):
  """[T,257] float32 like BFMNet + the photo's coefficients (infer_bfmvid.py:221-229): identity/texture/pose constant over
  the clip, expression varying; and the jittered angles of render_face (:84-90)."""
  rng = np.random.default_rng(seed)
  base = np.zeros(257, np.float32)
  base[:80] = rng.normal(size=80)
  base[144:224] = rng.normal(size=80)
  base[224:227] = rng.normal(0, 0.1, 3)
  base[227:254] = rng.normal(0, 0.15, 27)
  base[254:] = [0.02, -0.03, 0.1]
  coeff = np.tile(base, (frames, 1))
  coeff[:, 80:144] = rng.normal(0, 0.8, size=(frames, 64))
  angles = np.cumsum(np.full((frames, 3), 0.005, np.float32), 0).astype(np.float32)
  return coeff.astype(np.float32), angles


def rotation_matrices(angles):
  """Compute_rotation_matrix (reconstruct_mesh.py:68-93) for [T,3] angles: (Rz Ry Rx)^T per frame, float64 [T,3,3]."""
  out = []
  for ax, ay, az in np.asarray(angles):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    out.append((rz @ ry @ rx).T)
  return np.stack(out)


SH_A = (np.pi, 2 * np.pi / np.sqrt(3.0), 2 * np.pi / np.sqrt(8.0))
SH_C = (1 / np.sqrt(4 * np.pi), np.sqrt(3.0) / np.sqrt(4 * np.pi), 3 * np.sqrt(5.0) / np.sqrt(12 * np.pi))


def reconstruction_rotation(coeff, fm, angles, focal=1015.0, center=112.0):
  """Batched Reconstruction_rotation (reconstruct_mesh.py:198-223).  coeff [T,257], angles [T,3] ->
  dict(face_shape [T,N,3] (rotated), face_texture, face_color, face_projection [T,N,2], z_buffer [T,N,1], landmarks_2d [T,68,2])."""
  coeff = np.asarray(coeff)
  T = coeff.shape[0]
  idc, exc, texc, gamma, trans = coeff[:, :80], coeff[:, 80:144], coeff[:, 144:224], coeff[:, 227:254], coeff[:, 254:]   # :5-13
  shape = (idc @ fm.idBase.T + exc @ fm.exBase.T + fm.meanshape).reshape(T, -1, 3)                   # :21-25
  shape = shape - fm.meanshape.reshape(1, -1, 3).mean(axis=1, keepdims=True)                         # :27
  tex = (texc @ fm.texBase.T + fm.meantex).reshape(T, -1, 3)                                         # :59-60
  tri = (fm.tri - 1).astype(np.int32)
  pb = (fm.point_buf - 1).astype(np.int32)
  v1, v2, v3 = shape[:, tri[:, 0]], shape[:, tri[:, 1]], shape[:, tri[:, 2]]
  fn = np.cross(v1 - v2, v2 - v3)                                                                    # :43-46
  fn = np.concatenate([fn, np.zeros((T, 1, 3))], axis=1)
  vn = fn[:, pb].sum(axis=2)                                                                         # :50
  vn = vn / np.linalg.norm(vn, axis=2)[..., None]
  R = rotation_matrices(angles)
  vn_r = vn @ R                                                                                      # :208
  shape_r = shape @ R                                                                                # :211
  # Projection_layer (:100-122) applies `rotation` a SECOND time to the already rotated shape (:214 passes both)
  cam = (shape_r @ R + trans.reshape(T, 1, 3)) * np.array([1.0, 1.0, -1.0]) + np.array([0.0, 0.0, 10.0])
  aug = np.stack([focal * cam[..., 0] + center * cam[..., 2], focal * cam[..., 1] + center * cam[..., 2], cam[..., 2]], -1)
  proj = aug[..., :2] / aug[..., 2:3]
  zbuf = -aug[..., 2:3]
  proj = np.stack([proj[..., 0], 224 - proj[..., 1]], axis=2)                                        # :215
  g = gamma.reshape(T, 3, 9).astype(np.float64) + np.array([0.8, 0, 0, 0, 0, 0, 0, 0, 0])            # :133-135
  a0, a1, a2 = SH_A
  c0, c1, c2 = SH_C
  nx, ny, nz = vn_r[..., 0], vn_r[..., 1], vn_r[..., 2]
  Y = np.stack([np.full_like(nx, a0 * c0), -a1 * c1 * ny, a1 * c1 * nz, -a1 * c1 * nx, a2 * c2 * nx * ny, -a2 * c2 * ny * nz,
                a2 * c2 * 0.5 / np.sqrt(3.0) * (3 * np.square(nz) - 1), -a2 * c2 * nx * nz,
                a2 * c2 * 0.5 * (np.square(nx) - np.square(ny))], axis=2)                            # :145-155
  lit = np.einsum("tnk,tck->tnc", Y, g)
  color = lit * tex                                                                                  # :165-166
  return {"face_shape": shape_r, "face_texture": tex, "face_color": color, "face_projection": proj, "z_buffer": zbuf,
          "landmarks_2d": proj[:, fm.keypoints]}


def pack_for_raster(out):
  """infer_bfmvid.py:92-99: vertices = [x, 224-y, z_buffer] float32, colours clip(0,255) -> int32 -> float32."""
  vertices = np.concatenate([out["face_projection"], out["z_buffer"]], axis=2).astype(np.float32)
  colors = np.clip(out["face_color"], 0, 255).astype(np.int32).astype(np.float32)
  return vertices, colors
