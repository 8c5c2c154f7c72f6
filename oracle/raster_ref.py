"""Oracle for the flat-shaded z-buffer rasteriser (the "next" row of SURVEY.md 8f-1).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Two checkers:
  * render_colors_py : plain numpy/Python restatement of `_render_colors_core`
                       (utils/cython/mesh_core.cpp:169-231, point-in-triangle test :23-50), float32 arithmetic in the
                       reference's evaluation order so that it is bit-exact with the C++;
  * render_colors_ref: the reference's OWN mesh_core.cpp compiled into oracle/_ref/libmesh_core_ref.so by oracle/Makefile
                       (kind "reference").  PINNED: tests/test_raster.py checks the restatement against it bit for bit and
                       tests/golden/raster.npz holds outputs it produced in the build container.
"""
import ctypes
import os

import numpy as np

F = np.float32
_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libmesh_core_ref.so")


def _in_tri(px, py, p0, p1, p2):
  """isPointInTri, mesh_core.cpp:23-50, in float32 with the same operation order."""
  v0x, v0y = F(p2[0] - p0[0]), F(p2[1] - p0[1])
  v1x, v1y = F(p1[0] - p0[0]), F(p1[1] - p0[1])
  v2x, v2y = (px - p0[0]).astype(F), (py - p0[1]).astype(F)
  dot00 = F(F(v0x * v0x) + F(v0y * v0y))
  dot01 = F(F(v0x * v1x) + F(v0y * v1y))
  dot02 = (v0x * v2x).astype(F) + (v0y * v2y).astype(F)
  dot11 = F(F(v1x * v1x) + F(v1y * v1y))
  dot12 = (v1x * v2x).astype(F) + (v1y * v2y).astype(F)
  den = F(F(dot00 * dot11) - F(dot01 * dot01))
  inv = F(0) if den == 0 else F(F(1) / den)
  u = ((dot11 * dot02).astype(F) - (dot01 * dot12).astype(F)).astype(F) * inv
  v = ((dot00 * dot12).astype(F) - (dot01 * dot02).astype(F)).astype(F) * inv
  return (u >= 0) & (v >= 0) & ((u + v).astype(F) < 1)


def render_colors_py(vertices, triangles, colors, h, w, c=3, depth_init=-99999.0):
  """Returns (image uint8 [h,w,c], face_mask uint8 [h,w], depth float32 [h,w]) like the in-place C++ on zeroed buffers."""
  vertices = np.asarray(vertices, F).reshape(-1, 3)
  colors = np.asarray(colors, F).reshape(-1, c)
  tri = np.asarray(triangles, np.int32).reshape(-1, 3)
  image = np.zeros((h, w, c), np.uint8)
  mask = np.zeros((h, w), np.uint8)
  depth = np.full((h, w), depth_init, F)
  for t in range(tri.shape[0]):
    p0, p1, p2 = vertices[tri[t, 0]], vertices[tri[t, 1]], vertices[tri[t, 2]]
    x_min = max(int(np.ceil(min(p0[0], p1[0], p2[0]))), 0)
    x_max = min(int(np.floor(max(p0[0], p1[0], p2[0]))), w - 1)
    y_min = max(int(np.ceil(min(p0[1], p1[1], p2[1]))), 0)
    y_max = min(int(np.floor(max(p0[1], p1[1], p2[1]))), h - 1)
    if x_max < x_min or y_max < y_min:
      continue
    pd = F(F(F(p0[2] + p1[2]) + p2[2]) / F(3))
    ys, xs = np.mgrid[y_min:y_max + 1, x_min:x_max + 1]
    hit = (pd > depth[ys, xs]) & _in_tri(xs.astype(F), ys.astype(F), p0, p1, p2)
    if not hit.any():
      continue
    csum = F(F(colors[tri[t, 0]] + colors[tri[t, 1]]) + colors[tri[t, 2]])          # float sum, then (int)(.)/3
    col = (csum.astype(np.int32) // 3 + ((csum.astype(np.int32) % 3 != 0) & (csum.astype(np.int32) < 0))).astype(np.int32)
    image[ys[hit], xs[hit]] = col.astype(np.uint8)
    mask[ys[hit], xs[hit]] = 255
    depth[ys[hit], xs[hit]] = pd
  return image, mask, depth


def have_compiled_reference():
  return os.path.exists(_REF)


def render_colors_ref(vertices, triangles, colors, h, w, c=3, depth_init=-99999.0):
  """The reference's compiled _render_colors_core (mesh_core.h:63), called exactly as infer_bfmvid.py:100-108 does."""
  lib = ctypes.CDLL(_REF)
  fn = getattr(lib, "_Z19_render_colors_corePhS_PfPiS0_S0_iiii")
  fn.restype = None
  vertices = np.ascontiguousarray(np.asarray(vertices, F).reshape(-1))
  colors = np.ascontiguousarray(np.asarray(colors, F).reshape(-1))
  tri = np.ascontiguousarray(np.asarray(triangles, np.int32).reshape(-1))
  image = np.zeros(h * w * c, np.uint8)
  mask = np.zeros(h * w, np.uint8)
  depth = np.full(h * w, depth_init, F)
  P = ctypes.c_void_p
  fn(P(image.ctypes.data), P(mask.ctypes.data), P(vertices.ctypes.data), P(tri.ctypes.data), P(colors.ctypes.data),
     P(depth.ctypes.data), ctypes.c_int(tri.size // 3), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(c))
  return image.reshape(h, w, c), mask.reshape(h, w), depth.reshape(h, w)


def synthetic_mesh(seed=0, nlat=40, nlon=60, h=224, w=224, extra=True):
  """A bumpy ellipsoid projected into an h x w image (thousands of small triangles, front and back faces overlapping),
  plus duplicates (depth ties), a degenerate triangle, triangles partly / fully outside the image."""
  rng = np.random.default_rng(seed)
  th = np.linspace(0.05, np.pi - 0.05, nlat)
  ph = np.linspace(0, 2 * np.pi, nlon, endpoint=False)
  T, Pp = np.meshgrid(th, ph, indexing="ij")
  r = 1 + 0.08 * rng.normal(size=T.shape)
  x = (w / 2 + 0.42 * w * r * np.sin(T) * np.cos(Pp)).reshape(-1)
  y = (h / 2 + 0.47 * h * r * np.cos(T)).reshape(-1)
  z = (60 * r * np.sin(T) * np.sin(Pp)).reshape(-1)
  verts = np.stack([x, y, z], 1).astype(F)
  tris = []
  for i in range(nlat - 1):
    for j in range(nlon):
      a, b = i * nlon + j, i * nlon + (j + 1) % nlon
      tris += [[a, b, a + nlon], [b, b + nlon, a + nlon]]
  tris = np.array(tris, np.int32)
  if extra:
    n0 = verts.shape[0]
    more = np.array([[10.5, 10.5, 500], [10.5, 10.5, 500], [10.5, 10.5, 500],        # degenerate (inverDeno = 0)
                     [-30, 20, 400], [40, -25, 400], [35, 60, 400],                   # partly outside
                     [-50, -50, 900], [-10, -60, 900], [-20, -5, 900],                # fully outside
                     [w - 20.0, h - 30.0, 300], [w + 25.0, h - 10.0, 300], [w - 5.0, h + 20.0, 300]], F)
    verts = np.concatenate([verts, more])
    tris = np.concatenate([tris, np.array([[n0, n0 + 1, n0 + 2], [n0 + 3, n0 + 4, n0 + 5], [n0 + 6, n0 + 7, n0 + 8],
                                           [n0 + 9, n0 + 10, n0 + 11]], np.int32),
                           tris[100:140], tris[100:140][:, [1, 2, 0]]])              # exact depth ties, other vertex order
  colors = np.clip(rng.uniform(-5, 260, size=(verts.shape[0], 3)), 0, 255).astype(np.int32).astype(F)   # infer_bfmvid.py:98
  return verts, tris, colors
