"""Rasteriser parity (SURVEY.md 8f-1 "next" row): vp_render_colors / utils.mesh_core vs the reference's mesh_core.cpp.

CPU: the numpy restatement against the reference-captured golden (tests/golden/raster.npz, produced by the compiled
     reference) and, when oracle/_ref is built, against the compiled reference itself on a fresh seed.
GPU: the HIP kernels against the golden and the restatement, BIT-EXACT (uint8 image / mask, float32 depth bits),
     single frame through the reference's positional signature and a batch of frames in one launch."""
import os

import numpy as np
import pytest

from oracle import raster_ref as rr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster.npz")


def _case(tag):
  g = np.load(GOLD)
  return [g[tag + "_" + k] for k in ("vertices", "triangles", "colors", "image", "mask", "depth")]


@pytest.mark.parametrize("tag", ["a", "b"])
def test_restatement_matches_reference_golden(tag):
  v, t, c, img, mask, depth = _case(tag)
  h, w = mask.shape
  i2, m2, d2 = rr.render_colors_py(v, t, c, h, w)
  assert np.array_equal(i2, img) and np.array_equal(m2, mask)
  assert np.array_equal(d2.view(np.uint32), depth.view(np.uint32))
  assert 0.2 < (mask > 0).mean() < 0.95            # the fixture really covers pixels, and leaves background


@pytest.mark.skipif(not rr.have_compiled_reference(), reason="oracle/_ref not built (make -C oracle; needs /root/reference)")
def test_restatement_matches_compiled_reference():
  v, t, c = rr.synthetic_mesh(seed=17, nlat=20, nlon=24, h=120, w=100)
  a = rr.render_colors_ref(v, t, c, 120, 100)
  b = rr.render_colors_py(v, t, c, 120, 100)
  for x, y in zip(a, b):
    assert np.array_equal(x, y)


def test_edge_cases_restatement():
  # no triangles; everything outside; tie keeps the FIRST triangle (strict >, mesh_core.cpp:211)
  v = np.array([[2, 2, 5], [12, 2, 5], [2, 12, 5], [-9, -9, 1], [-5, -9, 1], [-9, -5, 1]], np.float32)
  col = np.array([[10, 20, 30]] * 3 + [[200, 100, 50]] * 3, np.float32)
  img, mask, depth = rr.render_colors_py(v, np.zeros((0, 3), np.int32), col, 16, 16)
  assert not img.any() and not mask.any() and (depth == np.float32(-99999.0)).all()
  img, mask, _ = rr.render_colors_py(v, np.array([[3, 4, 5]], np.int32), col, 16, 16)
  assert not mask.any()
  col2 = np.concatenate([col[:3], col[3:]])
  v2 = np.concatenate([v[:3], v[:3]])
  img, mask, _ = rr.render_colors_py(v2, np.array([[0, 1, 2], [3, 4, 5]], np.int32), col2, 16, 16)
  assert mask.any() and (img[mask > 0] == np.array([10, 20, 30], np.uint8)).all()


# ------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_gpu_render_colors_core_signature(tag):
  from voicepuppet_amd.utils import mesh_core
  v, t, c, img, mask, depth = _case(tag)
  h, w = mask.shape
  new_image = np.zeros(h * w * 3, np.uint8)
  face_mask = np.zeros(h * w, np.uint8)
  depth_buffer = (np.zeros(h * w) - 99999.0).astype(np.float32)
  mesh_core.render_colors_core(new_image, face_mask, v.reshape(-1).copy(), t.reshape(-1).copy(), c.reshape(-1).copy(), depth_buffer,
                               t.shape[0], h, w, 3)
  assert np.array_equal(new_image.reshape(h, w, 3), img)
  assert np.array_equal(face_mask.reshape(h, w), mask)
  assert np.array_equal(depth_buffer.view(np.uint32).reshape(h, w), depth.view(np.uint32))


@pytest.mark.gpu
def test_gpu_batched_frames_and_edges():
  import torch
  from voicepuppet_amd.utils import mesh_core
  h, w, F = 224, 224, 5
  base_v, tri, base_c = rr.synthetic_mesh(seed=23, h=h, w=w)
  rng = np.random.default_rng(1)
  vs, cs, want = [], [], []
  for f in range(F):
    v = base_v.copy()
    v[:, :2] += rng.normal(0, 1.5, size=(1, 2)).astype(np.float32)             # head motion
    v[:, 2] *= np.float32(1 + 0.05 * f)
    c = np.roll(base_c, f, axis=0)
    vs.append(v); cs.append(c)
    want.append(rr.render_colors_py(v, tri, c, h, w))
  dev = "cuda"
  image = torch.zeros(F, h, w, 3, dtype=torch.uint8, device=dev)
  mask = torch.zeros(F, h, w, dtype=torch.uint8, device=dev)
  depth = torch.full((F, h, w), -99999.0, dtype=torch.float32, device=dev)
  mesh_core.render_colors(image, mask, torch.from_numpy(np.stack(vs)).to(dev), torch.from_numpy(tri).to(dev),
                          torch.from_numpy(np.stack(cs)).to(dev), depth)
  for f in range(F):
    assert np.array_equal(image[f].cpu().numpy(), want[f][0]), f
    assert np.array_equal(mask[f].cpu().numpy(), want[f][1]), f
    assert np.array_equal(depth[f].cpu().numpy().view(np.uint32), want[f][2].view(np.uint32)), f
  # second pass over the SAME buffers changes nothing (strict > against the depths already written)
  before = image.clone()
  mesh_core.render_colors(image, mask, torch.from_numpy(np.stack(vs)).to(dev), torch.from_numpy(tri).to(dev),
                          torch.from_numpy(np.stack(cs)).to(dev), depth)
  assert torch.equal(before, image)
  # no triangles: buffers untouched; triangle 0 alone must win over the cleared buffer
  img0 = torch.zeros(1, 16, 16, 3, dtype=torch.uint8, device=dev)
  m0 = torch.zeros(1, 16, 16, dtype=torch.uint8, device=dev)
  d0 = torch.full((1, 16, 16), -99999.0, device=dev)
  v = torch.tensor([[[2, 2, 5], [12, 2, 5], [2, 12, 5]]], dtype=torch.float32, device=dev)
  col = torch.tensor([[[10, 20, 30]] * 3], dtype=torch.float32, device=dev)
  mesh_core.render_colors(img0, m0, v, torch.zeros(0, 3, dtype=torch.int32, device=dev), col, d0)
  assert not m0.any() and not img0.any()
  mesh_core.render_colors(img0, m0, v, torch.tensor([[0, 1, 2]], dtype=torch.int32, device=dev), col, d0)
  ref = rr.render_colors_py(v[0].cpu().numpy(), np.array([[0, 1, 2]], np.int32), col[0].cpu().numpy(), 16, 16)
  assert np.array_equal(img0[0].cpu().numpy(), ref[0]) and np.array_equal(m0[0].cpu().numpy(), ref[1])
