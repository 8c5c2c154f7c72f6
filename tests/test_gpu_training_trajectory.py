"""-m gpu: does the bf16 path TRAIN like the float32 path?  

The reference's shipped weights are 10,000-iteration runs (train_pixrefer.py:134, infer_bfmvid.py:217-218); the element-wise gradient
comparison of the bf16 path against the float64 graph is conditioning-limited (0.37-0.42 rel-L2, tests/test_gpu_fullwidth.py), so the
end-to-end statement for the benchmark dtype is made here: ngf = ndf = 64, the four-sample fixture (tests/golden/full_width_n4.npz),
200 iterations of the reference's schedule (Adam(D) then Adam(G), lr = 3e-4 * 0.999^floor(global_step / 1000)) on the float32 engine
and on the bf16 engine from the SAME initial weights.  A GAN's per-step losses are chaotic once the discriminator has saturated, so the
comparison is on 10-step window means at steps 10 / 50 / 100 / 200, with the bands stated below, and on what the reference's training
is for: Gen_loss_L1 (L1 + matte + 1x perceptual, weight 500 in Gen_loss) must fall by the same factor on both paths and the trained
generators must produce the same pixels.  The two curves of this test are committed as profiles/r05_train_curves.json
(scripts/train_curves.py writes them)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import train_curves as tc

pytestmark = pytest.mark.gpu

STEPS = 200
# relative band of the bf16 window mean around the float32 window mean, per checkpoint step
# measured (profiles/r05_train_curves.json): Gen_loss_L1 0.4 / 0.0 / 2.8 / 2.4 %, Perceptual_loss <= 2 %, GAN terms <= 0.57 nats apart
BAND_L1 = {10: 0.05, 50: 0.05, 100: 0.08, 200: 0.08}          # Gen_loss_L1 and Perceptual_loss (smooth, monotone terms)
ABS_GAN = 1.2                                                   # Discrim_loss / Gen_loss_GAN window means: absolute band (nats)


def test_bf16_loss_trajectory_tracks_float32_over_200_steps():
  ngf, seed, batch = tc.fixture_batch()
  params = tc.initial_params(seed)
  f, pix_f = tc.run_curves("f32", STEPS, params, batch)
  b, pix_b = tc.run_curves("bf16", STEPS, params, batch)
  assert np.isfinite(f).all() and np.isfinite(b).all()
  K = {k: i for i, k in enumerate(tc.KEYS)}
  lines = []
  for at in (10, 50, 100, 200):
    wf, wb = tc.window_mean(f, min(at, STEPS - 1)), tc.window_mean(b, min(at, STEPS - 1))
    lines.append("step %3d  " % at + "  ".join("%s %.4f/%.4f" % (k, wf[i], wb[i]) for k, i in K.items()))
    for k in ("Gen_loss_L1", "Perceptual_loss"):
      rel = abs(wb[K[k]] - wf[K[k]]) / abs(wf[K[k]])
      assert rel < BAND_L1[at], (at, k, wf[K[k]], wb[K[k]])
    for k in ("Discrim_loss", "Gen_loss_GAN"):
      assert abs(wb[K[k]] - wf[K[k]]) < ABS_GAN, (at, k, wf[K[k]], wb[K[k]])
  print("\n" + "\n".join(lines))
  # the generator learns: Gen_loss_L1 falls, by the same factor on both paths
  fall_f = tc.window_mean(f, STEPS - 1)[K["Gen_loss_L1"]] / f[0, K["Gen_loss_L1"]]
  fall_b = tc.window_mean(b, STEPS - 1)[K["Gen_loss_L1"]] / b[0, K["Gen_loss_L1"]]
  print("Gen_loss_L1 after %d steps / at step 0: f32 %.4f, bf16 %.4f" % (STEPS, fall_f, fall_b))
  assert fall_f < 0.7 and fall_b < 0.7, (fall_f, fall_b)
  assert abs(fall_b - fall_f) < 0.1 * fall_f, (fall_f, fall_b)
  # ... and the two trained generators agree on the image (mean |Outputs - targets| within 10 % of each other)
  l1_f, l1_b = float(np.abs(pix_f - batch[2]).mean()), float(np.abs(pix_b - batch[2]).mean())
  print("mean |Outputs - targets| after training: f32 %.4f, bf16 %.4f" % (l1_f, l1_b))
  assert abs(l1_b - l1_f) < 0.1 * l1_f, (l1_f, l1_b)
