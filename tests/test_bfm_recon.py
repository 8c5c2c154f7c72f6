"""BFM reconstruction parity: vp_bfm_reconstruct / utils.reconstruct_mesh vs the reference's
utils/reconstruct_mesh.py.  The golden was captured from the reference module itself (tests/golden/make_golden.py).

Tolerances: the float64 intermediates agree to 1e-12 relative (summation order of the 144-term bases differs from numpy's
BLAS); the float32 vertices and integer colours handed to the rasteriser, and the rasterised uint8 frames, must be IDENTICAL."""
import os

import numpy as np
import pytest

from oracle import bfm_ref as br
from oracle import raster_ref as rr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bfm_recon.npz")
NAMES = ["face_shape", "face_texture", "face_color", "face_projection", "z_buffer", "landmarks_2d"]


def _golden():
  g = np.load(GOLD)
  fm = br.synthetic_facemodel(int(g["model_seed"]))
  chk = np.array([fm.idBase.sum(), fm.exBase.sum(), fm.texBase.sum(), fm.meanshape.sum(), fm.meantex.sum(), float(fm.tri.sum()),
                  float(fm.point_buf.sum()), float(fm.keypoints.sum())])
  assert np.array_equal(chk, g["model_checksum"]), "seeded face model differs from the one the golden was made with"
  return g, fm


def _rel(a, b):
  return np.abs(a - b).max() / np.abs(b).max()


def test_oracle_matches_reference_golden():
  g, fm = _golden()
  out = br.reconstruction_rotation(g["coeff"], fm, g["angles"])
  for n in NAMES:
    assert _rel(out[n], g[n]) < 1e-13, n
  v, c = br.pack_for_raster(out)
  assert np.array_equal(v, g["vertices"]) and np.array_equal(c, g["colors"])
  img, mask, _ = rr.render_colors_py(v[2], fm.tri - 1, c[2], 224, 224)
  assert np.array_equal(img, g["images"][2]) and np.array_equal(mask, g["masks"][2])


def test_float32_model_follows_reference_dtype_promotion():
  # with a float32 .mat the reference's einsum runs in float32: the float64 restatement stays within float32 rounding of it
  fm32, fm64 = br.synthetic_facemodel(9, dtype=np.float32), br.synthetic_facemodel(9)
  coeff, angles = br.synthetic_coeffs(2, 1)
  a, b = br.reconstruction_rotation(coeff, fm32, angles), br.reconstruction_rotation(coeff, fm64, angles)
  for n in NAMES:
    assert _rel(a[n], b[n]) < 5e-6, n


def test_host_rotation_matrices_bit_identical_to_per_frame_form():
  # the clip-wide numpy form the host wrapper uploads == the reference's one-frame-at-a-time construction (oracle restates it)
  from voicepuppet_amd.utils.reconstruct_mesh import Compute_rotation_matrix
  ang = np.random.default_rng(0).normal(0, 0.5, size=(300, 3)).astype(np.float32)
  assert np.array_equal(Compute_rotation_matrix(ang), br.rotation_matrices(ang))


@pytest.mark.gpu
def test_gpu_reconstruct_matches_reference_golden():
  from voicepuppet_amd.utils import reconstruct_mesh as vrm
  g, fm = _golden()
  model = vrm.DeviceFaceModel(fm)
  out = vrm.reconstruct_clip(g["coeff"], model, g["angles"])
  for n in NAMES:
    assert _rel(out[n].cpu().numpy(), g[n]) < 1e-12, n
  assert np.array_equal(out["vertices"].cpu().numpy(), g["vertices"])
  assert np.array_equal(out["colors"].cpu().numpy(), g["colors"])
  # constant texture coefficients over the clip (infer_bfmvid.py:226-229): computed once, same result
  o2 = vrm.reconstruct_clip(g["coeff"], model, g["angles"], shared_texture=True)
  assert np.array_equal(o2["colors"].cpu().numpy(), g["colors"]) and o2["face_texture"].shape[0] == 1


@pytest.mark.gpu
def test_gpu_reference_signature_and_long_clip():
  from voicepuppet_amd.utils import reconstruct_mesh as vrm
  g, fm = _golden()
  res = vrm.Reconstruction_rotation(g["coeff"][1:2], fm, g["angles"][1:2])          # one frame, numpy in / numpy out
  assert len(res) == 6
  for n, r in zip(NAMES, res):
    assert r.dtype == np.float64 and r.shape == g[n][1:2].shape and _rel(r, g[n][1:2]) < 1e-12, n
  # 70 frames: more than one 32-frame pass of the basis kernel, against the oracle
  coeff, angles = br.synthetic_coeffs(70, 8)
  want = br.reconstruction_rotation(coeff, fm, angles)
  out = vrm.reconstruct_clip(coeff, vrm.DeviceFaceModel(fm), angles)
  for n in NAMES:
    assert _rel(out[n].cpu().numpy(), want[n]) < 1e-12, n
  v, c = br.pack_for_raster(want)
  assert np.array_equal(out["vertices"].cpu().numpy(), v) and np.array_equal(out["colors"].cpu().numpy(), c)


@pytest.mark.gpu
def test_gpu_clip_renderer_matches_reference_frames():
  from voicepuppet_amd.utils import reconstruct_mesh as vrm
  g, fm = _golden()
  image, mask = vrm.ClipRenderer(fm)(g["coeff"], g["angles"])
  assert np.array_equal(image.cpu().numpy(), g["images"])
  assert np.array_equal(mask.cpu().numpy(), g["masks"])
