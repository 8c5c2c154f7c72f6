"""-m gpu: does the bf16 path TRAIN like the float32 path?  (re-stated in round 6 per VERDICT r5 item 6.)

The reference's shipped weights are 10,000-iteration runs (train_pixrefer.py:134, infer_bfmvid.py:217-218); the element-wise gradient
comparison of the bf16 path against the float64 graph is conditioning-limited (0.37-0.42 rel-L2, tests/test_gpu_fullwidth.py), so the
end-to-end statement for the benchmark dtype is made here: ngf = ndf = 64, the four-sample fixture (tests/golden/full_width_n4.npz),
200 iterations of the reference's schedule (Adam(D) then Adam(G), lr = 3e-4 * 0.999^floor(global_step / 1000)).

Round 5 ran ONE seed and compared 10-step window means with the float32 ones inside hard-coded bands (5-8 %); changing one kernel
selection knob moved the step-50 perceptual term 8.9 % - and in round 6 changing the summation order of ONE batch-norm reduction moved
the same build's step-50 Gen_loss_L1 from 5 % to 10 % off float32: per-step losses of this GAN carry 10-20 % bumps that last 2-3
iterations and arrive at different iterations in every run, so a 10-step window at the knee of the curve measures whether it holds a
bump.  Now (scripts/train_spread.py; curves and statistics in profiles/r06_train_spread.json / .txt):
  * THREE seeds of the initial weights, each run on the float32 engine, on the float32 engine from weights perturbed by 1e-6 relative
    (the float32 path's own chaos), on the bf16 engine, and on the bf16 engine under the plan heuristics of rounds 2-5 (vp_tune "patch_min_blocks"
    384 + "igemm_splitk_target" 128: other kernel classes / K splits; round 5's version of this arm left its band);
  * 50-iteration windows [25, 75), [75, 125), [125, 175) - geometric means of the decaying terms, arithmetic means (nats) of the GAN terms;
  * the yardstick is the float32 path itself, measured in this run and not written down: Gen_loss_L1 / Gen_loss of every bf16 run within
    the largest SEED-TO-SEED spread of the float32 runs (measured 5.7 % / 4.9 %; bf16 sits 0.8-2.3 % from the float32 run of its own
    seed across two builds of round 6, a 1e-6 perturbation of the float32 weights moves float32 by 0.2-2.0 %); the two GAN terms within 1.65 x the largest float32
    seed-to-seed range (the 95 % band of the difference between two float32 initialisations estimated from a three-sample range: 0.60 /
    0.55 nats; bf16 0.05-0.29 nats, perturbed float32 0.07-0.22); the perceptual term's level is a property of the initial weights (seed
    spread 55-67 %: no yardstick), so it gets a stated band of 12 % (bf16 3.8-6.6 %, perturbed float32 1.5-2.1 %).
Gen_loss_L1 (L1 + matte + 1x perceptual, weight 500 in Gen_loss) must fall by the same factor on both paths and the trained generators
must produce the same pixels.  (These runs use the SHIPPED plan options - the first layers write no raw output, store_first_raw = 0:
ADVICE r5 asked for a whole-step test on that default path.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import train_curves as tc
import train_spread as ts

pytestmark = pytest.mark.gpu

STEPS = 200
SEEDS = (0, 1, 2)
BAND_PERCEPTUAL = 0.12
GAN_RANGE_FACTOR = 1.65


def test_bf16_loss_trajectory_sits_inside_the_float32_seed_spread():
  runs = ts.run_all(STEPS, SEEDS)
  for c in runs.values():
    assert np.isfinite(c).all()
  dist = ts.distances(runs, SEEDS, STEPS)
  print("\n" + ts.report(dist))
  K = {k: i for i, k in enumerate(tc.KEYS)}
  band = {}
  for k in tc.KEYS:
    spread = max(dist[at][k]["f32_seed_spread"] for at in ts.AT)
    band[k] = BAND_PERCEPTUAL if k == "Perceptual_loss" else (spread if k in ts.REL_KEYS else GAN_RANGE_FACTOR * spread)
  print("bands: " + ", ".join("%s %.4f" % kv for kv in band.items()))
  assert band["Gen_loss_L1"] < 0.10 and band["Gen_loss"] < 0.10, band        # (the yardstick itself stays a yardstick)
  for at in ts.AT:
    for k, i in K.items():
      rel = k in ts.REL_KEYS
      for s in SEEDS:
        wf = ts.window_stat(runs[(s, "f32")][:, i], at, rel)
        for arm in ("bf16", "bf16_alt"):
          wb = ts.window_stat(runs[(s, arm)][:, i], at, rel)
          d = abs(wb - wf) / (abs(wf) if rel else 1.0)
          assert d <= band[k], (at, k, s, arm, wf, wb, d, band[k])
  # the generator learns: Gen_loss_L1 falls, by the same factor on both paths, for every seed and both bf16 arms
  i = K["Gen_loss_L1"]
  for s in SEEDS:
    f = runs[(s, "f32")]
    fall_f = ts.window_stat(f[:, i], STEPS - ts.HALF, True) / f[0, i]
    assert fall_f < 0.7, (s, fall_f)
    for arm in ("bf16", "bf16_alt"):
      b = runs[(s, arm)]
      fall_b = ts.window_stat(b[:, i], STEPS - ts.HALF, True) / b[0, i]
      assert abs(fall_b - fall_f) < 0.1 * fall_f, (s, arm, fall_f, fall_b)


def test_trained_generators_agree_on_the_image():
  """... and the generators the two paths train produce the same pixels (mean |Outputs - targets| within 10 % of each other after 200
  iterations from the fixture's seed)."""
  ngf, seed, batch = tc.fixture_batch()
  params = tc.initial_params(seed)
  f, pix_f = tc.run_curves("f32", STEPS, params, batch)
  b, pix_b = tc.run_curves("bf16", STEPS, params, batch)
  l1_f, l1_b = float(np.abs(pix_f - batch[2]).mean()), float(np.abs(pix_b - batch[2]).mean())
  print("mean |Outputs - targets| after training: f32 %.4f, bf16 %.4f" % (l1_f, l1_b))
  assert abs(l1_b - l1_f) < 0.1 * l1_f, (l1_f, l1_b)
