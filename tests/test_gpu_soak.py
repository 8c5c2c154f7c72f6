"""-m gpu: the lock-step soak comparisons of scripts/soak*.py in short form (the soak is what found the batch-8
kernel-family bug, so it belongs in the suite).  Full width (ngf = ndf = 64), bf16, FRESH random batches every step, several steps so
that Adam state, re-packed weights and saturating losses are in play:
  * the three-stream schedule vs the single-stream schedule: gradient and parameter arenas bit-identical after every step, at the
    batch sizes where the planner switches kernel families (4, 8, 16 frames; 32 is covered by the benchmark run itself);
  * the fused backward + Adam + re-pack call vs the separate calls: parameters bit-identical after every step."""
import pytest
import torch

from voicepuppet_amd import _lib
from voicepuppet_amd.engine import PixReferEngine

pytestmark = pytest.mark.gpu


def _first_difference(a, b):
  for name, x, y in (("grads_g", a.grads_g, b.grads_g), ("grads_d", a.grads_d, b.grads_d), ("params_g", a.params_g, b.params_g),
                     ("params_d", a.params_d, b.params_d)):
    if not torch.equal(x, y):
      idx = int((x != y).nonzero().flatten()[0])
      man = a.manifests[0 if name.endswith("_g") else 1]
      which = [vn for vn, off, shape in man if off <= idx < off + int(torch.tensor(shape).prod())]
      return "%s differs, first in %s (max |diff| %.3e)" % (name, which[:1], float((x - y).abs().max()))
  return None


@pytest.mark.parametrize("n,steps", [(4, 6), (8, 6), (16, 4)])
def test_overlapped_and_single_stream_schedules_stay_bit_identical(n, steps):
  L = _lib.lib()
  dev = torch.device("cuda", 0)
  a = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  b = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  p = a.random_params(seed=0)
  a.load_params(p)
  b.load_params(p)
  g = torch.Generator(device=dev).manual_seed(1)
  b.set_option("overlap", 0)              # per handle (vp_pixrefer_set_option): `a` keeps the multi-stream schedule
  for s in range(steps):
    batch = [torch.rand(n, 256, 256, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
    a.train_step(*batch, lr=3e-4)
    b.train_step(*batch, lr=3e-4)
    torch.cuda.synchronize()
    diff = _first_difference(a, b)
    assert diff is None, "step %d: %s" % (s, diff)
    assert torch.isfinite(a.params_g).all() and torch.isfinite(a.params_d).all()


def test_fused_update_and_separate_calls_stay_bit_identical():
  dev = torch.device("cuda", 0)
  n = 8
  a = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  b = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  p = a.random_params(seed=0)
  a.load_params(p)
  b.load_params(p)
  b.fused_update = False
  g = torch.Generator(device=dev).manual_seed(2)
  for s in range(6):
    batch = [torch.rand(n, 256, 256, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
    a.train_step(*batch, lr=3e-4)
    b.train_step(*batch, lr=3e-4)
    torch.cuda.synchronize()
    assert torch.equal(a.params_g, b.params_g) and torch.equal(a.params_d, b.params_d), "step %d" % s


def test_engine_created_without_its_own_streams_matches():
  """vp_pixrefer_desc::streams = 1: no side / branch streams exist at all.  The fused backward + update step must run there - it once recorded an event on the missing side stream - and
  give the parameters of the default three-stream engine bit for bit."""
  import numpy as np
  dev = torch.device("cuda", 0)
  n, ngf = 2, 8
  rng = np.random.default_rng(4)
  batches = [[torch.tensor(rng.uniform(size=(n, 256, 256, c)).astype(np.float32), device=dev) for c in (6, 6, 3, 3)] for _ in range(3)]
  a = PixReferEngine(n, 256, ngf, ngf, dtype="bf16", training=True)
  p = a.random_params(seed=5)
  a.load_params(p)
  b = PixReferEngine(n, 256, ngf, ngf, dtype="bf16", training=True, streams=1)
  b.load_params(p)
  for batch in batches:
    a.train_step(*batch, lr=3e-4)
    b.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  assert torch.equal(a.params_g, b.params_g) and torch.equal(a.params_d, b.params_d)
