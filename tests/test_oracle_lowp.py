"""The device-dataflow oracle (oracle/pixrefer_lowp_ref.py) with q = identity must equal the plain float64 oracle:
that pins the restated dataflow (materialised activations, grouped BN, accumulation order) independently of rounding."""
import numpy as np
import pytest

from oracle import pixrefer_lowp_ref as lowp
from oracle import pixrefer_ref as ref


@pytest.mark.slow
def test_identity_rounding_reproduces_float64_oracle():
  ngf = ndf = 4
  rng = np.random.default_rng(3)
  p = ref.init_params(ngf, ndf, seed=2, dtype=np.float32)
  for k in p:
    if k.endswith('beta') or ('layer_1/' in k and k.endswith('bias')) or ('encoder_1/' in k and k.endswith('bias')):
      p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
  p = {k: v.astype(np.float64) for k, v in p.items()}
  batch = [rng.uniform(size=(2, 256, 256, c)).astype(np.float32).astype(np.float64) for c in (6, 6, 3, 3)]
  a = ref.forward_backward(p, *batch, ngf=ngf, ndf=ndf)
  b = lowp.forward_backward(p, *batch, ngf=ngf, ndf=ndf, q=lowp.IDENT)
  for k in ('Discrim_loss', 'Gen_loss_GAN', 'Gen_loss_L1', 'Gen_loss', 'Perceptual_loss'):
    assert b[k] == pytest.approx(a[k], rel=1e-6), k
  np.testing.assert_allclose(b['Outputs_raw'], a['Outputs_raw'], rtol=1e-5, atol=1e-6)
  for key in ('Discrim_grads', 'Gen_grads'):
    for name, g in a[key].items():
      if np.all(g == 0):
        continue
      err = np.linalg.norm(b[key][name] - g) / np.linalg.norm(g)
      # the only non-f64 pieces with q = identity are the f32-rounded scale/shift/statistics (device convention)
      assert err < 2e-4, (name, err)


def test_round_bf16_is_nearest_even():
  x = np.array([1.0, 1.00390625, 1.01171875, -3.1415926, 65504.0, 1e-40], dtype=np.float32)
  r = lowp.round_bf16(x)
  assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.015625      # tie -> even mantissa; above tie -> up
  assert abs(r[3] + 3.140625) < 1e-12 and np.isfinite(r).all()
