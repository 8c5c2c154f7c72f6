"""-m gpu: every kernel class the benchmark plans run is covered by an oracle op case.

The planner picks a kernel family / tile / split-K by grid size (plan_pixrefer.hip plan_net, conv_ops.h), so "the op tests pass"
only says something about the step if the op cases land on the SAME kernel classes the step launches (the batch-8 bug
the soak found was two kernel families with different K-sum orders).  This test builds the three plans BASELINE.json names
(global batch 32 at 256x256, batch 8 at 512x512, and the 4-per-GPU share of strong scaling), collects the class of every conv-like
launch of one step from the executor's own profile records (vp_profile_collect: family / variant, operand type, tile - the same
names bench.py reports), and checks each against the classes the parity cases of test_gpu_ops.py run on.  A class that only exists
inside the step (f32-output kernels of decoder_1, tap-GEMM helpers) must be listed in IN_STEP with the step-parity test that covers it.
"""
import ctypes
import json

import numpy as np
import pytest
import torch

from voicepuppet_amd import _lib
from voicepuppet_amd.engine import PixReferEngine

import gpu_util as gu
import test_gpu_ops as T

pytestmark = pytest.mark.gpu

# classes that cannot be reached through the single-op entry points, with the in-situ oracle test that covers them
IN_STEP = {
    "wgrad_tr_bf16_128x128": "decoder_1 weight gradient: 8-channel dY against the virtual concat of two 64-channel tensors (an operand tile "
                             "straddles the concat: generic loader); tests/test_gpu_step.py::test_step_parity[bf16] checks decoder_1/ tightly",
    "cout1bwd_bf16_512x16": "layer_5 backward-data (512 <- 1 channel, with layer_4's batch-norm backward sums): "
                            "tests/test_gpu_step.py::test_one_output_channel_backward_kernel_in_situ against the generic kernels, test_gpu_fullwidth.py against the oracle",
    "cout1wgrad_bf16_16x512": "layer_5 weight gradient (a wave per pixel, 16 x 8 sums per lane): "
                              "tests/test_gpu_step.py::test_one_output_channel_backward_kernel_in_situ against the generic kernels, test_gpu_fullwidth.py against the oracle",
    "cout4_bf16_16x16": "decoder_1 forward (128 -> 4 channels, f32 output): tests/test_gpu_step.py::test_step_parity[bf16], "
                        "test_thin_decoder_tile_kernel_in_situ, test_gpu_fullwidth.py",
}


def _collect():
  L = _lib.lib()
  n = L.vp_profile_collect(None, 0)
  buf = ctypes.create_string_buffer(int(n) + 16)
  L.vp_profile_collect(buf, len(buf))
  return {r["name"] for r in json.loads(buf.value.decode())}


def plan_classes(batch, height, dtype):
  eng = PixReferEngine(batch, height, 64, 64, dtype=dtype, training=True)
  eng.load_params(eng.random_params(seed=0))
  g = torch.Generator(device="cpu").manual_seed(1)
  b = [torch.rand(batch, height, height, c, generator=g).cuda() for c in (6, 6, 3, 3)]
  eng.train_step(*b, lr=3e-4)
  torch.cuda.synchronize()
  eng.profile(1)
  try:
    eng.train_step(*b, lr=3e-4)
    torch.cuda.synchronize()
    return _collect()
  finally:
    eng.profile(0)


def op_case_classes(dtype, min_blocks, old_wgrad_split=False):
  """Classes the op cases of test_gpu_ops.py run on (device side only: their numerics are asserted there).  old_wgrad_split: the weight-
  gradient K split of rounds 2-5 (no per-tile slab cost: 64 and more slabs on small cases), which is what puts an op-sized case on the
  256 x 256 tile of wgrad_tr.hip - a batch-32 plan reaches it with its pixel counts (tests/test_gpu_ops.py::test_weight_gradient_on_the_256x256_tile)."""
  L = _lib.lib()
  L.vp_tune(b"patch_min_blocks", min_blocks)
  L.vp_tune(b"wgrad_slab_tile_x1000", 0 if old_wgrad_split else -1)
  L.vp_profile_enable(1)
  try:
    for case in T.FWD_CASES + EXTRA_CASES:
      kind, n, h, w, cin, cout, k, s, p, in_act, out_act, affine = case
      x, wt, b, sc, sh = T.make_case(case)
      d = gu.conv_desc(kind, n, h, w, cin, cout, k, s, p, dtype, in_act, out_act)
      gu.conv_fwd(d, x, sc, sh, wt, b, dtype)
      if cout >= 8 and (cout & (cout - 1)) == 0:
        ho, wo = gu.out_hw(d)
        dy = np.zeros((n, ho, wo, cout))
        d0 = gu.conv_desc(kind, n, h, w, cin, cout, k, s, p, dtype)
        gu.conv_bwd_data(d0, dy, wt, dtype)
        gu.conv_bwd_weight(gu.conv_desc(kind, n, h, w, cin, cout, k, s, p, dtype, in_act), x, sc, sh, dy, wt.shape, dtype)
    torch.cuda.synchronize()
    return _collect()
  finally:
    L.vp_profile_enable(0)
    L.vp_tune(b"patch_min_blocks", -1)       # (< 0: the library default)
    L.vp_tune(b"wgrad_slab_tile_x1000", -1)


# op cases (numerics asserted by tests/test_gpu_ops.py::test_more_kernel_classes) that exist to put a benchmark class under the oracle
EXTRA_CASES = [c for _, c in T.CLASS_CASES]


@pytest.mark.parametrize("batch,height", [(32, 256), (8, 512), (4, 256)])
def test_every_benchmark_kernel_class_has_an_oracle_op_case(batch, height):
  want = plan_classes(batch, height, "bf16")
  have = op_case_classes("bf16", 1) | op_case_classes("bf16", 384) | op_case_classes("bf16", 384, old_wgrad_split=True)
  missing = sorted(c for c in want if c not in have and c not in IN_STEP)
  print("\n[batch %d, %dx%d] step classes: %s" % (batch, height, height, sorted(want)))
  assert not missing, "kernel classes of the bs-%d/%d^2 plan without an oracle op case: %s" % (batch, height, missing)


def test_f32_parity_path_classes_are_covered():
  """The float32 plan (the path that meets the 1e-3 pixel tolerance) at the 4-per-GPU batch."""
  want = plan_classes(4, 256, "f32")
  have = op_case_classes("f32", 1) | op_case_classes("f32", 384) | op_case_classes("f32", 384, old_wgrad_split=True)
  in_step = {k.replace("bf16", "f32") for k in IN_STEP}
  missing = sorted(c for c in want if c not in have and c not in in_step)
  assert not missing, missing
