"""-m gpu: the whole PixReferNet G+D step (vp_pixrefer_*) against the numpy oracle on the same seeded
inputs and parameters.  f32 path: <= 1e-3 relative L2 on generator pixels, losses <= 1e-4 relative,
discriminator gradients <= 1e-4 (BASELINE.md 2.3 / SURVEY.md 8d).  Generator gradients pass through the
VGG trunk's ~3e7 ReLU masks and the sign() of the L1 loss: a float32 forward flips a handful of masks
that a float64 forward does not, so ANY float32 implementation differs from the float64 oracle by a few
1e-3 there (the numpy oracle itself run in float32: 2.6e-3 on d(VGG input), 3.7e-3..5.5e-3 on generator
kernels; the HIP f32 path measures 1.2e-3..1.6e-3).  Tolerance for those tensors: 5e-3.
bf16 path: reported, expected ~1e-2 on pixels (bf16 storage, f32 accumulate)."""
import numpy as np
import pytest
import torch

from oracle import pixrefer_ref as ref
from voicepuppet_amd import _lib
from voicepuppet_amd.engine import PixReferEngine

import gpu_util as gu

pytestmark = pytest.mark.gpu


def synth(n, h, seed):
  rng = np.random.default_rng(seed)
  f = lambda c: rng.uniform(size=(n, h, h, c)).astype(np.float32)
  return f(6), f(6), f(3), f(3)


def make_params(ngf, ndf, seed):
  p = ref.init_params(ngf, ndf, seed=seed, dtype=np.float32)
  rng = np.random.default_rng(seed + 1)
  for k in p:
    if k.endswith("beta") or (k.endswith("bias") and ("encoder_1/" in k or "encoder_fg_1/" in k or "decoder_1/" in k or "layer_1/" in k or "layer_5/" in k)):
      p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
  return p


@pytest.fixture(scope="module")
def oracle_step():
  ngf = ndf = 8
  n, h = 2, 256
  p = make_params(ngf, ndf, 3)
  batch = synth(n, h, 11)
  p64 = {k: v.astype(np.float64) for k, v in p.items()}
  st = ref.TrainState(p64, ngf, ndf)
  nodes = st.step(*[b.astype(np.float64) for b in batch])
  return dict(ngf=ngf, ndf=ndf, n=n, h=h, params=p, batch=batch, nodes=nodes, after=st.p)


def run_engine(o, dtype):
  eng = PixReferEngine(o["n"], o["h"], o["ngf"], o["ndf"], dtype=dtype, training=True)
  eng.load_params(o["params"])
  eng.set_option("store_first_raw", 1)          # the parity tests read every layer's stored output (a step skips the first layers' raw ones)
  dev = [torch.tensor(b, device="cuda") for b in o["batch"]]
  eng.forward(*dev)
  eng.backward()
  torch.cuda.synchronize()
  return eng


def test_manifest_matches_oracle(oracle_step):
  o = oracle_step
  eng = PixReferEngine(1, 256, o["ngf"], o["ndf"], dtype="f32", training=True)
  g, d = ref.param_manifest(o["ngf"], o["ndf"])
  assert [(n, s) for n, _, s in eng.manifests[0]] == [(n, tuple(s)) for n, s in g]
  assert [(n, s) for n, _, s in eng.manifests[1]] == [(n, tuple(s)) for n, s in d]
  assert [(n, s) for n, _, s in eng.manifests[2]] == [(n, tuple(s)) for n, s in ref.vgg_manifest()]


@pytest.mark.parametrize("dtype,tol_pix,tol_grad,tol_loss", [("f32", 1e-3, 5e-3, 1e-4), ("bf16", 3e-2, 1.5e-1, 3e-2)])
def test_step_parity(oracle_step, dtype, tol_pix, tol_grad, tol_loss):
  o = oracle_step
  nodes = o["nodes"]
  eng = run_engine(o, dtype)
  got = eng.losses()
  report = {}
  for k in ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss"):
    report[k] = abs(got[k] - nodes[k]) / abs(nodes[k])
  pix = gu.rel_l2(eng.tensor("Outputs_raw").cpu().numpy(), nodes["Outputs_raw"])
  fg = gu.rel_l2(eng.tensor("Outputs_FG").cpu().numpy(), nodes["Outputs_FG"])
  worst = {}
  for which, key in ((1, "Discrim_grads"), (0, "Gen_grads")):
    grads = eng.get_params(which, src=eng.grads_d if which == 1 else eng.grads_g)
    for name, g in grads.items():
      r = nodes[key][name]
      if np.all(r == 0):
        assert np.all(g == 0), name
        continue
      worst[name] = gu.rel_l2(g, r)
  print("\n[%s] loss rel err %s\n pixels relL2 %.3e fg %.3e\n worst grads %s" % (
      dtype, {k: "%.2e" % v for k, v in report.items()}, pix, fg,
      sorted(((v, k) for k, v in worst.items()), reverse=True)[:5]))
  assert pix < tol_pix and fg < tol_pix
  assert max(report.values()) < tol_loss, report
  if dtype == "bf16":
    # bf16 activations flip ~0.3 % of the piecewise-linear masks per layer relative to a float64 forward
    # (4e-3 relative forward error x density of pre-activations at 0), i.e. ~5 % gradient noise per layer
    # on this random-noise batch; only tensors within two layers of a loss are compared tightly here.
    near = ("decoder_1/", "merged2_decoder_2/", "layer_5/", "layer_4/")
    worst = {k: v for k, v in worst.items() if any(t in k for t in near)}
  bad = {k: v for k, v in worst.items() if v > tol_grad}
  assert not bad, bad
  if dtype == "f32":
    bad_d = {k: v for k, v in worst.items() if k.startswith("discriminator") and v > 1e-4}
    assert not bad_d, bad_d


def test_adam_update_and_determinism(oracle_step):
  o = oracle_step
  eng = run_engine(o, "f32")
  g1 = eng.grads_g.clone()
  d1 = eng.grads_d.clone()
  eng.adam_step(3e-4)
  torch.cuda.synchronize()
  after = o["after"]
  for which, key in ((0, "Gen_grads"), (1, "Discrim_grads")):
    got = eng.get_params(which)
    for name, v in got.items():
      # first Adam step moves every weight by ~lr_t; compare the update, not the value
      before = o["params"][name].astype(np.float64)
      du, dr = v - before, after[name] - before
      if np.abs(dr).max() == 0:
        assert np.abs(du).max() == 0, name
        continue
      # sign(g) * lr_t update: an element whose gradient is smaller than the float32 path's gradient tolerance (5e-3 of the tensor,
      # see the module docstring) may legitimately take the other sign, and ONE such flip in a 64-element beta is 0.25 in rel-L2;
      # compare the elements whose oracle gradient is above that noise floor, and all elements loosely
      gr = np.abs(o["nodes"][key][name])
      firm = gr > 2e-2 * gr.max()
      assert gu.rel_l2(du[firm], dr[firm]) < 2e-2, (name, gu.rel_l2(du[firm], dr[firm]))
      assert gu.rel_l2(du, dr) < 3e-1, (name, gu.rel_l2(du, dr))
  # bit-reproducible: a second engine on the same data gives identical gradients
  eng2 = run_engine(o, "f32")
  assert torch.equal(eng2.grads_g, g1) and torch.equal(eng2.grads_d, d1)


def test_inference_plan_matches_training_forward(oracle_step):
  o = oracle_step
  eng = PixReferEngine(o["n"], o["h"], o["ngf"], o["ndf"], dtype="f32", training=False)
  eng.load_params(o["params"])
  dev = [torch.tensor(b, device="cuda") for b in o["batch"]]
  eng.forward(dev[0], dev[1][..., :3].contiguous(), dev[2])
  torch.cuda.synchronize()
  out = ref.inference({k: v.astype(np.float64) for k, v in o["params"].items()},
                      *[b.astype(np.float64) for b in (o["batch"][0], o["batch"][1][..., :3], o["batch"][2])], ngf=o["ngf"])
  got = (eng.tensor("Outputs_raw").cpu().numpy() + 1) / 2
  assert gu.rel_l2(got, out["Outputs"]) < 1e-3


def test_sample22_fixture_forward_and_step_vs_golden():
  """BASELINE config 1: generator forward on the reference's own sample/22.jpg (decoded fixture), plus one G+D
  step, against the golden values the float64 oracle produced in the build container."""
  import os
  G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
  s, d = np.load(os.path.join(G, "sample22_256.npz")), np.load(os.path.join(G, "mini_step.npz"))
  frame, face3d, matte, bg = [s[k].astype(np.float32) / 255.0 for k in ("frame", "face3d", "matte", "background")]
  inputs = np.concatenate([face3d, face3d], axis=-1)[None]
  fg = np.concatenate([frame * matte, frame * matte], axis=-1)[None]
  ngf = int(d["ngf"])
  p = ref.init_params(ngf, ngf, seed=int(d["seed"]), dtype=np.float32)
  dev = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
  inf = PixReferEngine(1, 256, ngf, ngf, dtype="f32", training=False)
  inf.load_params(p)
  inf.forward(dev(inputs), dev(fg[..., :3]), dev(bg[None]))
  out = (inf.tensor("Outputs_raw").cpu().numpy() + 1) / 2
  assert gu.rel_l2(out[0, 96:160, 96:160], d["Infer_Outputs_crop"]) < 1e-3
  np.testing.assert_allclose(out.mean(axis=(0, 1, 2)), d["Infer_Outputs_mean"], rtol=1e-4)
  eng = PixReferEngine(1, 256, ngf, ngf, dtype="f32", training=True)
  eng.load_params(p)
  eng.forward(dev(inputs), dev(fg), dev(frame[None]), dev(matte[None]))
  eng.backward()
  got = eng.losses()
  for k in ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss"):
    assert got[k] == pytest.approx(float(d[k]), rel=1e-4), k
  assert gu.rel_l2((eng.tensor("Outputs_raw").cpu().numpy()[0, 96:160, 96:160] + 1) / 2, d["Outputs_crop"]) < 1e-3
  names = list(d["grad_names"])
  gg, gd = eng.get_params(0, src=eng.grads_g), eng.get_params(1, src=eng.grads_d)
  norms = np.array([np.linalg.norm(gg[n] if n.startswith("generator") else gd[n]) for n in names])
  ok = d["grad_norms"] > 0
  # N=1: batch-norm over a single 1x1 bottleneck pixel kills the gradient of the deepest layers (norm ~ 0); compare the rest
  big = ok & (d["grad_norms"] > 1e-6 * d["grad_norms"].max())
  np.testing.assert_allclose(norms[big], d["grad_norms"][big], rtol=2e-2)


def test_step_parity_bf16_against_rounding_aware_oracle(oracle_step):
  """bf16 device path vs the device-dataflow oracle with bf16 rounding at every storage point
  (oracle/pixrefer_lowp_ref.py): ReLU / leaky-ReLU masks agree by construction, so gradients are compared tightly.
  The generator's N=2 bottleneck batch-norm turns one-ulp differences into sign flips (conditioning of the test problem,
  not of the kernels), so the oracle is teacher-forced: every generator layer is recomputed from the device's stored
  tensors of the layers before it (checked per layer), and the backward then runs on identical activations."""
  from oracle import pixrefer_lowp_ref as lowp
  o = oracle_step
  eng = run_engine(o, "bf16")
  out4 = eng.tensor("gen_out4").cpu().numpy()
  p64 = {k: v.astype(np.float64) for k, v in o["params"].items()}
  g_dev = {sc: eng.tensor("g/" + sc).float().cpu().numpy() for sc, *_ in ref.generator_spec(o["ngf"])}
  d_dev = {sc: eng.tensor("d/" + sc).float().cpu().numpy() for sc, *_ in ref.discriminator_spec(o["ndf"])}
  # the few-pixel batch-normalised tensors are float32 on the device (round 4): the oracle rounds them to float32 as well
  hi = {sc for sc, *_ in ref.generator_spec(o["ngf"]) if sc != "decoder_1" and eng.tensor("g/" + sc).dtype == torch.float32}
  print("\n[float32 few-pixel tensors]", sorted(hi))
  assert "merged_encoder_5" in hi and "merged_decoder_5" in hi and "encoder_2" not in hi
  nodes = lowp.forward_backward(p64, *[b.astype(np.float64) for b in o["batch"]], ngf=o["ngf"], ndf=o["ndf"], q=lowp.round_bf16,
                                out4_override=out4, g_override=g_dev, d_override=d_dev, hi=hi)
  # layer-by-layer forward parity of the generator: each layer recomputed from the device's own previous tensors
  fwd = nodes["G"].fwd_err
  print("\n[bf16 generator forward, per layer from device inputs] worst:", sorted(((v, k) for k, v in fwd.items()), reverse=True)[:4])
  assert max(fwd.values()) < 4e-3, fwd
  # ... and of the discriminator (its three applications as one batch of 3N)
  fwd_d = nodes["D"].fwd_err
  print("[bf16 discriminator forward, per layer from device inputs] worst:", sorted(((v, k) for k, v in fwd_d.items()), reverse=True)[:4])
  assert max(fwd_d.values()) < 4e-3, fwd_d
  got = eng.losses()
  for k in ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss"):
    assert got[k] == pytest.approx(nodes[k], rel=1e-4), k
  assert gu.rel_l2(eng.tensor("Outputs_raw").cpu().numpy(), nodes["Outputs_raw"]) < 1e-5
  T = lambda name: eng.tensor(name).float().cpu().numpy()
  mid = {"d_din": gu.rel_l2(T("d_din")[..., 3:6], nodes["d_din"]), "d_vin": gu.rel_l2(T("d_vin")[..., :3], nodes["d_vin"]),
         "dy4": gu.rel_l2(T("d_gen_out4")[..., :4], nodes["dy4"])}
  worst = {}
  for which, key in ((1, "Discrim_grads"), (0, "Gen_grads")):
    grads = eng.get_params(which, src=eng.grads_d if which == 1 else eng.grads_g)
    for name, g in grads.items():
      r = nodes[key][name]
      if np.all(r == 0):
        continue
      worst[name] = gu.rel_l2(g, r)
  top = sorted(((v, k) for k, v in worst.items()), reverse=True)[:6]
  print("\n[bf16 vs rounding-aware oracle] intermediates %s\n worst gradient rel-L2: %s" % (mid, top))
  # residual: f32 accumulation order differs from the oracle's, which flips the bf16 rounding (1 ulp = 0.4 %) of a few
  # per cent of the elements at each of the ~8 backward stages
  # (before the discriminator was teacher-forced too, d_din of this 8-channel mini net moved between 1.5e-2 and 3.7e-2 with anything
  # that changes one bf16 rounding upstream - the split-K setting of a layer, the summation order of the generator's bottleneck)
  assert max(mid.values()) < 3e-2, mid
  # merged_encoder_5's output is 1x1: at this fixture's N = 2 its batch-norm is over TWO values per channel, whose backward pass
  # vanishes analytically - what reaches the layer's kernel gradient is rounding residue, and since round 4 the device forms it from
  # float32 tensors in float32 arithmetic where the oracle rounds float64 results to float32 (measured 5.8e-2; the well-posed case,
  # N = 4 at full width, is test_gpu_fullwidth.py::test_bf16_generator_gradients_at_full_width_against_the_rounding_aware_oracle: 5e-3)
  bound = lambda k: 1.5e-1 if "/merged_encoder_5/" in k else 5e-2
  bad = {k: v for k, v in worst.items() if v > bound(k)}   # measured worst elsewhere: 4.2e-2 (a batch-norm gamma), typical 1e-2
  assert not bad, bad


@pytest.mark.gpu
def test_staged_generator_backward_equals_monolithic():
  """backward_g in its three all-reduce stages == the single call, bit for bit; the buckets tile the gradient arena."""
  import torch
  from voicepuppet_amd.engine import PixReferEngine
  ngf = ndf = 8
  p = ref.init_params(ngf, ndf, seed=3, dtype=np.float32)
  rng = np.random.default_rng(5)
  batch = [torch.tensor(rng.uniform(size=(2, 256, 256, c)).astype(np.float32), device="cuda") for c in (6, 6, 3, 3)]
  eng = PixReferEngine(2, 256, ngf, ndf, dtype="bf16", training=True)
  eng.load_params(p)
  eng.forward(*batch); eng.backward_d(); eng.backward_g()
  torch.cuda.synchronize()
  want = eng.grads_g.clone()
  eng.grads_g.fill_(float("nan"))
  eng.forward(*batch); eng.backward_d()
  buckets = eng.grad_buckets_g()
  assert buckets[0][1] == eng.grads_g.numel() and buckets[2][0] == 0
  assert buckets[0][0] == buckets[1][1] and buckets[1][0] == buckets[2][1] and buckets[2][1] > 0
  for stage, (lo, hi) in enumerate(buckets):
    eng.backward_g_stage(stage)
    torch.cuda.synchronize()
    assert torch.equal(eng.grads_g[lo:hi], want[lo:hi]), stage        # this bucket is final after its stage
  assert torch.equal(eng.grads_g, want)


@pytest.mark.gpu
def test_decoder_1_four_channel_kernel_in_situ():
  """decoder_1 at a width where the dedicated 4-channel transposed-conv kernel runs (Cin = 64 = 2 MFMA steps per tap):
  its f32 output against the float64 deconvolution of the SAME bf16 inputs the device fed it (teacher forcing)."""
  from oracle import nn_ops as ops
  ngf = 32
  p = ref.init_params(ngf, ngf, seed=11, dtype=np.float32)
  rng = np.random.default_rng(2)
  inputs, fg, tgt = [torch.tensor(rng.uniform(size=(1, 256, 256, c)).astype(np.float32), device="cuda") for c in (6, 3, 3)]
  eng = PixReferEngine(1, 256, ngf, ngf, dtype="bf16", training=False)
  eng.load_params(p)
  eng.set_option("store_first_raw", 1)          # encoder_1's raw output is read below (a step does not store it)
  eng.forward(inputs, fg, tgt)
  torch.cuda.synchronize()
  c2 = eng.tensor("g/merged2_decoder_2").float()
  sc, sh = eng.tensor("g/merged2_decoder_2:scale").view(-1), eng.tensor("g/merged2_decoder_2:shift").view(-1)
  x_c2 = torch.relu(torch.addcmul(sh, sc, c2)).to(torch.bfloat16).float()          # act_apply: relu(fma(scale, y, shift)) -> bf16
  x_e1 = torch.relu(eng.tensor("g/encoder_1").float())
  x = torch.cat([x_c2, x_e1], dim=-1).cpu().numpy().astype(np.float64)
  w = gu.rounded(p["generator/decoder_1/conv2d_transpose/kernel"], "bf16")
  want = ops.deconv4s2_fwd(x, w, p["generator/decoder_1/conv2d_transpose/bias"].astype(np.float64))
  got = eng.tensor("g/decoder_1").cpu().numpy()
  assert got.shape == want.shape == (1, 256, 256, 4)
  assert gu.rel_l2(got, want) < 2e-3, gu.rel_l2(got, want)       # fma-vs-float64 activation rounding flips a few bf16 ulps


@pytest.mark.gpu
def test_conv1_2_register_resident_weights_kernel_in_situ():
  """VGG conv1_2 (64 -> 64 channels, 3x3) runs on conv_c64.hip in every bf16 training plan: forward + bias + relu, the fused 2x2 max
  pool (`pool1`), and the backward-data pass with the relu'(conv1_1 output) product - each against the float64 convolution of the SAME
  bf16 tensors the device fed the kernel (teacher forcing; bound 4e-3 = bf16 rounding of the stored result), and against the unrolled
  patch kernel the layer ran on before (vp_tune("c64", 0): different K-sum order, same rounding points)."""
  from oracle import nn_ops as ops
  from voicepuppet_amd import _lib
  L = _lib.lib()
  n = 2
  got = {}
  for on in (1, 0):
    L.vp_tune(b"c64", on)
    try:
      eng = PixReferEngine(n, 256, 8, 8, dtype="bf16", training=True)
      eng.load_params(eng.random_params(5))
      g = torch.Generator(device="cpu").manual_seed(9)
      batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
      eng.profile(1)
      eng.forward(*batch); eng.backward()
      torch.cuda.synchronize()
      classes = {r["name"] for r in eng.profile_collect()}
      eng.profile(0)
      assert any(c.startswith("c64_") for c in classes) == bool(on), classes
      got[on] = {k: eng.tensor(k).float().cpu().numpy() for k in ("v/conv1/conv1_1", "v/conv1/conv1_2", "v/pool1", "v/conv1/conv1_2:dy", "v/conv1/conv1_1:dy",
                                                                  "v/conv2/conv2_1", "v/conv2/conv2_2", "v/pool2", "v/conv2/conv2_1:dy", "v/pool1:dy")}
      if on:
        w = gu.rounded(eng.get_params(2)["vgg_16/conv1/conv1_2/weights"], "bf16")
        b = eng.get_params(2)["vgg_16/conv1/conv1_2/biases"].astype(np.float64)
      del eng
    finally:
      L.vp_tune(b"c64", 1)
  t = got[1]
  x = t["v/conv1/conv1_1"].astype(np.float64)                          # [2N] relu outputs, as stored (bf16)
  want = ops.relu(ops.conv2d_fwd(x[n:], w, b, 1, 1))                   # the fake half (the profiled step runs on one stream and stores both)
  assert gu.rel_l2(t["v/conv1/conv1_2"][n:], want) < 4e-3, gu.rel_l2(t["v/conv1/conv1_2"][n:], want)
  y = t["v/conv1/conv1_2"]
  pool = y.reshape(2 * n, 128, 2, 128, 2, 64).max(axis=(2, 4))
  assert np.array_equal(t["v/pool1"], pool)                            # the fused pool is the pool of the stored tensor, bit for bit
  # backward-data (fake half only: dy tensors hold N images): dX = conv_bwd(dY, W) * relu'(conv1_1 output of the fake half)
  dy = t["v/conv1/conv1_2:dy"].astype(np.float64)
  dx = ops.conv2d_bwd(x[n:n + 1], w, dy[:1], 1, 1, need_dw=False)[0] * (x[n:n + 1] > 0)
  assert gu.rel_l2(t["v/conv1/conv1_1:dy"][:1], dx) < 4e-3, gu.rel_l2(t["v/conv1/conv1_1:dy"][:1], dx)
  # (the gradient tensors of the two runs also differ by what a handful of flipped bf16 roundings of the forward output do to the pool's
  # argmax and the relu masks upstream: bound 3e-2 there, 3e-3 on the forward tensors)
  # the 128-channel layers of the same kernel family (conv2_1 forward: four waves of 32 channels; conv2_2 both passes and conv2_1
  # backward-data: waves of 16 channels, 128 input channels): the fused pool again bit for bit, everything against the patch kernels
  y2 = t["v/conv2/conv2_2"]
  assert np.array_equal(t["v/pool2"], y2.reshape(2 * n, 64, 2, 64, 2, 128).max(axis=(2, 4)))
  for k, tol in (("v/conv1/conv1_2", 3e-3), ("v/pool1", 3e-3), ("v/conv1/conv1_1:dy", 3e-2), ("v/conv2/conv2_1", 3e-3), ("v/conv2/conv2_2", 4e-3),
                 ("v/pool2", 4e-3), ("v/conv2/conv2_1:dy", 3e-2), ("v/pool1:dy", 3e-2)):
    assert gu.rel_l2(got[1][k], got[0][k]) < tol, (k, gu.rel_l2(got[1][k], got[0][k]))


@pytest.mark.gpu
def test_transposed_conv_classes_with_register_resident_weights_in_situ():
  """The backward-data passes of the 64 -> 128 stride-2 convolutions (discriminator layer_2 in both gradient passes, encoder_fg_2,
  and encoder_2, which ADDS to the gradient decoder_1 wrote into encoder_1's buffer first) run on conv_dc64.hip at ngf = ndf = 64: the
  gradient tensors they write - the lrelu'(reference) product included - against the unrolled patch kernel they ran on before
  (vp_tune("dc64", 0): different K-sum order, same rounding points), and the weight gradients downstream of them."""
  from voicepuppet_amd import _lib
  L = _lib.lib()
  n = 2
  got, grads = {}, {}
  for on in (1, 0):
    L.vp_tune(b"dc64", on)
    try:
      eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
      eng.load_params(eng.random_params(5))
      g = torch.Generator(device="cpu").manual_seed(9)
      batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
      eng.profile(1)
      eng.forward(*batch); eng.backward()
      torch.cuda.synchronize()
      classes = {r["name"] for r in eng.profile_collect()}
      eng.profile(0)
      assert any(c.startswith("dc64_") for c in classes) == bool(on), classes
      got[on] = {k: eng.tensor(k).float().cpu().numpy() for k in ("d/layer_1:dy", "g/encoder_fg_1:dy", "g/encoder_1:dy", "d_din")}
      grads[on] = (eng.grads_d.clone(), eng.grads_g.clone())
      del eng
    finally:
      L.vp_tune(b"dc64", 1)
  for k in got[1]:
    assert np.isfinite(got[1][k]).all() and np.abs(got[1][k]).max() > 0, k
    assert gu.rel_l2(got[1][k], got[0][k]) < 4e-3, (k, gu.rel_l2(got[1][k], got[0][k]))
  rel = lambda x, y: float((x - y).norm() / y.norm())
  assert rel(grads[1][0], grads[0][0]) < 4e-3 and rel(grads[1][1], grads[0][1]) < 4e-3, (rel(grads[1][0], grads[0][0]), rel(grads[1][1], grads[0][1]))


@pytest.mark.gpu
def test_stride2_convs_with_register_resident_weights_and_block_statistics_in_situ():
  """The 64 -> 128 stride-2 convolutions in front of a batch-norm (encoder_2, encoder_fg_2, discriminator layer_2 with its three
  batch-norm groups) on conv_s2c64.hip (forced below its size threshold: vp_tune("s2c64", 1)): raw outputs, and the batch-norm scale /
  shift / mean / rstd formed from the per-block partial rows - group boundaries inside a block's tile walk, blocks that never see a
  group - against the generic kernel + its per-tile statistics (vp_tune("s2c64", 0)), and the gradients downstream."""
  from voicepuppet_amd import _lib
  L = _lib.lib()
  n = 2
  got, grads = {}, {}
  names = [net + f for net in ("g/encoder_2", "g/encoder_fg_2", "d/layer_2") for f in ("", ":scale", ":shift", ":mean", ":rstd")]
  for on in (1, 0):
    L.vp_tune(b"s2c64", on)
    try:
      eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
      eng.load_params(eng.random_params(5))
      g = torch.Generator(device="cpu").manual_seed(9)
      batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
      eng.profile(1)
      eng.forward(*batch); eng.backward()
      torch.cuda.synchronize()
      classes = {r["name"] for r in eng.profile_collect()}
      eng.profile(0)
      assert any(c.startswith("s2c64_") for c in classes) == bool(on), classes
      got[on] = {k: eng.tensor(k).float().cpu().numpy() for k in names}
      grads[on] = (eng.grads_d.clone(), eng.grads_g.clone())
      del eng
    finally:
      L.vp_tune(b"s2c64", 512)
  for k in names:
    assert np.isfinite(got[1][k]).all() and np.abs(got[1][k]).max() > 0, k
    tol = 4e-3 if ":" not in k else 1e-3
    assert gu.rel_l2(got[1][k], got[0][k]) < tol, (k, gu.rel_l2(got[1][k], got[0][k]))
  rel = lambda x, y: float((x - y).norm() / y.norm())
  # a different K-sum order in three FORWARD layers moves every rounding downstream: the generator's bf16 gradients answer a rounding-level
  # change of the activations with a few per cent (they sit 0.3-0.4 from the float64 graph whatever is stored how: EXPERIMENTS.md 0.1 of
  # round 4; measured here 0.058), the discriminator's with < 1 %
  assert rel(grads[1][0], grads[0][0]) < 2e-2 and rel(grads[1][1], grads[0][1]) < 0.15, (rel(grads[1][0], grads[0][0]), rel(grads[1][1], grads[0][1]))


@pytest.mark.gpu
def test_two_output_backward_of_the_last_wide_decoder_with_register_resident_weights_in_situ():
  """merged2_decoder_2 (256 -> 64 transposed convolution over [decoder | encoder_2 skip]): both data gradients in ONE launch of
  conv_s2c64.hip's two-output form (even / odd blocks = the two 128-channel halves; relu'(reference) on packed bf16; the second half can ADD to a
  gradient another consumer wrote first) against the generic two-output GEMM (vp_tune("s2c64_pair", 0): same forward, different K-sum
  order in this one launch), on the gradient tensors it writes and on everything downstream."""
  from voicepuppet_amd import _lib
  L = _lib.lib()
  n = 2
  got, grads, calls = {}, {}, {}
  names = ["g/merged2_decoder_3:dy", "g/encoder_2:dy", "g/encoder_1:dy", "g/merged2_decoder_4:dy"]
  L.vp_tune(b"s2c64", 1)
  try:
    for on in (1, 0):
      L.vp_tune(b"s2c64_pair", on)
      eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
      eng.load_params(eng.random_params(5))
      g = torch.Generator(device="cpu").manual_seed(9)
      batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
      eng.profile(1)
      eng.forward(*batch); eng.backward()
      torch.cuda.synchronize()
      recs = eng.profile_collect()
      eng.profile(0)
      calls[on] = sum(int(r["calls"]) for r in recs if r["name"].startswith("s2c64_"))
      got[on] = {k: eng.tensor(k).float().cpu().numpy() for k in names}
      grads[on] = eng.grads_g.clone()
      del eng
  finally:
    L.vp_tune(b"s2c64_pair", 1)
    L.vp_tune(b"s2c64", 512)
  assert calls[1] == calls[0] + 1 and calls[0] >= 3, calls        # the three forward layers + this launch
  for k in names:
    assert np.isfinite(got[1][k]).all() and np.abs(got[1][k]).max() > 0, k
    # the launch's own first output tightly; tensors that collect further contributions downstream of it (and pass through their
    # batch-norm backward at N = 2) answer its rounding-level differences with up to 1 % (measured 9e-3 on encoder_2)
    tol = 6e-3 if k == names[0] else 2e-2
    assert gu.rel_l2(got[1][k], got[0][k]) < tol, (k, gu.rel_l2(got[1][k], got[0][k]))
  rel = float((grads[1] - grads[0]).norm() / grads[0].norm())
  assert rel < 2e-2, rel


@pytest.mark.gpu
def test_last_wide_decoder_forward_with_register_resident_weights_in_situ():
  """merged2_decoder_2 forward (256 -> 64 transposed convolution over the virtual concat of two 128-channel tensors, batch-norm behind it)
  on conv_dc64.hip's conv_dc256_kernel (from 8 frames up, where the layer is on the parity-class patch plan): raw output and the batch-norm scale / shift / mean / rstd formed from its
  per-block partial rows against the unrolled patch kernel + its per-tile statistics (vp_tune("dc64", 0), which also moves the 128 -> 64
  backward classes back: forward tensors only are compared)."""
  from voicepuppet_amd import _lib
  L = _lib.lib()
  n = 8
  got = {}
  names = ["g/merged2_decoder_2" + f for f in ("", ":scale", ":shift", ":mean", ":rstd")] + ["Outputs_raw"]
  for on in (1, 0):
    L.vp_tune(b"dc64", on)
    try:
      eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
      eng.load_params(eng.random_params(5))
      g = torch.Generator(device="cpu").manual_seed(9)
      batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
      eng.profile(1)
      eng.forward(*batch)
      torch.cuda.synchronize()
      classes = {r["name"] for r in eng.profile_collect()}
      eng.profile(0)
      assert any(c.startswith("dc256_") for c in classes) == bool(on), classes
      got[on] = {k: eng.tensor(k).float().cpu().numpy() for k in names}
      del eng
    finally:
      L.vp_tune(b"dc64", 1)
  for k in names:
    assert np.isfinite(got[1][k]).all() and np.abs(got[1][k]).max() > 0, k
    tol = 1e-3 if ":" in k else (4e-3 if k.startswith("g/") else 2e-2)
    assert gu.rel_l2(got[1][k], got[0][k]) < tol, (k, gu.rel_l2(got[1][k], got[0][k]))


@pytest.mark.gpu
def test_first_layers_store_their_raw_output_only_on_request():
  """encoder_1 / encoder_fg_1 / layer_1 (no batch-norm) write the activations their consumers read from the conv epilogue; the raw output
  is not stored in a step (nobody reads it): vp_pixrefer_tensor refuses it, everything downstream is bit-identical with and without
  vp_pixrefer_set_option("store_first_raw", 1), and with the option the stored tensor is what the activations were formed from."""
  n = 2
  out = {}
  for keep in (0, 1):
    eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)      # (64 output channels: the width the direct first-layer kernel takes)
    eng.load_params(eng.random_params(3))
    if keep:
      eng.set_option("store_first_raw", 1)
    g = torch.Generator(device="cpu").manual_seed(4)
    batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
    eng.forward(*batch); eng.backward()
    torch.cuda.synchronize()
    out[keep] = (eng.tensor("Outputs_raw").clone(), eng.tensor("losses").clone(), eng.grads_g.clone(), eng.grads_d.clone())
    if keep:
      for name in ("g/encoder_1", "g/encoder_fg_1", "d/layer_1"):
        y = eng.tensor(name).float()
        assert torch.isfinite(y).all() and y.abs().max() > 0
    else:
      for name in ("g/encoder_1", "g/encoder_fg_1", "d/layer_1"):
        with pytest.raises(RuntimeError, match="store_first_raw"):
          eng.tensor(name)
      eng.tensor("g/encoder_2")                 # (batch-normalised layers are stored as ever)
    del eng
  for a, b in zip(out[0], out[1]):
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_forward_is_hipgraph_capturable():
  """The library's launch sequence is fixed, allocates nothing and never synchronises: one inference forward captured
  into a hipGraph replays to the identical output (the claim of DESIGN.md section 2)."""
  ngf = 8
  eng = PixReferEngine(1, 256, ngf, ngf, dtype="bf16", training=False)
  eng.load_params(eng.random_params(4))
  rng = np.random.default_rng(6)
  x = [torch.tensor(rng.uniform(size=(1, 256, 256, c)).astype(np.float32), device="cuda") for c in (6, 3, 3)]
  eng.forward(*x)                      # first call packs the weights; later calls are pure kernel launches
  torch.cuda.synchronize()
  want = eng.tensor("Outputs_raw").clone()
  g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
      eng.forward(*x)
  eng.tensor("Outputs_raw").zero_()
  g.replay()
  torch.cuda.synchronize()
  assert torch.equal(eng.tensor("Outputs_raw"), want)


@pytest.mark.gpu
@pytest.mark.parametrize("n,h", [(4, 256), (1, 512)])
def test_full_width_bf16_against_f32_path(n, h):
  """ngf = ndf = 64 (the benchmark width: 128x256 / 256x256 tiles, wave-specialised kernels, tap-GEMM, epilogue statistics all
  engage), batch 4: the bf16 step against the f32 step of the same engine - two different instantiations of every kernel.
  The oracle is far too slow at this width; the f32 path is pinned to it at the mini sizes above."""
  res = {}
  for dt in ("f32", "bf16"):
    eng = PixReferEngine(n, h, 64, 64, dtype=dt, training=True)
    eng.load_params(eng.random_params(21))
    g = torch.Generator(device="cpu").manual_seed(3)
    batch = [torch.rand(n, h, h, c, generator=g).cuda() for c in (6, 6, 3, 3)]
    eng.forward(*batch); eng.backward()
    torch.cuda.synchronize()
    res[dt] = dict(out=eng.tensor("Outputs_raw").clone(), losses=eng.losses(), gd=eng.grads_d.clone(), gg=eng.grads_g.clone())
    assert all(np.isfinite(v) for v in res[dt]["losses"].values())
    del eng
    torch.cuda.empty_cache()
  a, b = res["f32"], res["bf16"]
  rel = lambda x, y: float((x - y).norm() / y.norm())
  assert rel(b["out"], a["out"]) < 1e-2                       # measured 3.3e-3
  for k in ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Perceptual_loss"):
    assert abs(b["losses"][k] - a["losses"][k]) <= 1e-2 * abs(a["losses"][k]), (k, a["losses"][k], b["losses"][k])   # measured <= 2.4e-3
  # whole-arena gradient agreement (bf16 mask-flip noise is per element; the arena norm of the difference stays small)
  assert rel(b["gd"], a["gd"]) < 0.15, rel(b["gd"], a["gd"])     # measured 8.0e-2
  assert rel(b["gg"], a["gg"]) < 0.05, rel(b["gg"], a["gg"])     # measured 1.0e-2


@pytest.mark.gpu
def test_step_parity_512_f32():
  """BASELINE configs 4/5 geometry (512x512): D layer_4 is 63x63 and layer_5 62x62 (odd, non-power-of-two grids in the tap-GEMM
  and the padded-grid weight-gradient walk), the bottleneck reaches 2x2.  f32 engine vs the float64 oracle, N = 1."""
  ngf = ndf = 8
  n, h = 1, 512
  p = make_params(ngf, ndf, 5)
  batch = synth(n, h, 13)
  nodes = ref.forward_backward({k: v.astype(np.float64) for k, v in p.items()}, *[b.astype(np.float64) for b in batch], ngf=ngf, ndf=ndf)
  eng = run_engine(dict(n=n, h=h, ngf=ngf, ndf=ndf, params=p, batch=batch), "f32")
  got = eng.losses()
  for k in ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss"):
    assert abs(got[k] - nodes[k]) <= 1e-4 * abs(nodes[k]), (k, got[k], nodes[k])
  assert gu.rel_l2(eng.tensor("Outputs_raw").cpu().numpy(), nodes["Outputs_raw"]) < 1e-3
  for which, key, tol in ((1, "Discrim_grads", 1e-4), (0, "Gen_grads", 5e-3)):
    grads = eng.get_params(which, src=eng.grads_d if which == 1 else eng.grads_g)
    for name, g in grads.items():
      r = nodes[key][name]
      if np.all(r == 0):
        assert np.all(g == 0), name
      else:
        assert gu.rel_l2(g, r) < tol, (name, gu.rel_l2(g, r))


@pytest.mark.gpu
def test_bucketed_train_step_through_rccl_single_rank():
  """engine.train_step's data-parallel branch (staged generator backward, four RCCL all-reduces on arena slices issued on the
  communication stream behind explicit events: parallel.GradExchange) on a one-rank NCCL group, which really executes the
  collectives: f32 transport (ncclAvg in place) must equal the plain step bit for bit; bf16 transport (pack -> bf16 sum -> unpack)
  must equal the plain step with every gradient rounded to bf16 once, i.e. parameters within a few Adam sign flips of near-zero
  gradients and gradients within 2^-8 relative.  (Multi-rank equality: gloo on CPU and tests/test_gpu_dp.py.)"""
  import os
  import socket
  import torch.distributed as dist
  if dist.is_initialized():
    pytest.skip("a process group already exists in this process")
  s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
  try:
    ngf = 8
    rng = np.random.default_rng(9)
    batch = [torch.tensor(rng.uniform(size=(2, 256, 256, c)).astype(np.float32), device="cuda") for c in (6, 6, 3, 3)]
    outs = {}
    for mode in ("plain", "f32", "bf16"):
      eng = PixReferEngine(2, 256, ngf, ngf, dtype="bf16", training=True)
      eng.load_params(eng.random_params(2))
      if mode != "plain":
        eng.grad_transport = mode
      for k in range(2):
        if mode != "plain" and k == 1:
          eng._exchange.timing = True             # round 4: HIP events on the communication stream around every bucket (bench.py `distributed.buckets`)
        eng.train_step(*batch, lr=3e-4, group=None if mode == "plain" else dist.group.WORLD)
      torch.cuda.synchronize()
      if mode != "plain":
        bk = eng._exchange.bucket_ms()
        assert [b["name"] for b in bk] == ["discriminator", "generator stage 0", "generator stage 1", "generator stage 2"]      # (round 6: the discriminator's bucket right behind its loss pass)
        nel = [eng.grads_d.numel()] + [hi - lo for lo, hi in eng.grad_buckets_g()]
        assert [b["bytes"] for b in bk] == [n * (2 if mode == "bf16" else 4) for n in nel]
        assert all(b["allreduce_ms"] > 0 and b["update_ms"] > 0 for b in bk) and bk[0]["wait_ms"] is None and all(b["wait_ms"] >= 0 for b in bk[1:])
      outs[mode] = (eng.params_g.clone(), eng.params_d.clone(), eng.grads_g.clone(), eng.grads_d.clone())
    for a, b in zip(outs["plain"], outs["f32"]):
      assert torch.equal(a, b)
    # bf16 transport: the gradients of the second step differ by one bf16 rounding (plus what the slightly different first update did)
    for k in (2, 3):
      ref_g, got_g = outs["plain"][k].double(), outs["bf16"][k].double()
      assert float((got_g - ref_g).norm() / ref_g.norm()) < 2e-2
    for k in (0, 1):
      # two Adam steps of +-lr each: parameters stay within 2 lr of each other, and on average far closer (sign flips are rare)
      d = (outs["plain"][k] - outs["bf16"][k]).abs()
      assert float(d.max()) <= 4 * 3e-4 * 1.01 and float(d.mean()) < 0.05 * 3e-4
    # bench.py's multi-rank leg (watchdog beats, per-bucket timing into the `distributed` record) on this one-rank group: the 8-GPU
    # scaling run is the driver's, so the code it will execute runs here first (world = 2 is only the number the record is scaled by)
    import bench
    res = bench.run_config(2, 256, "bf16", 2, 1, 0, 2, torch.device("cuda", 0), dist.group.WORLD, False, "bf16")
    assert res["ms_per_step"] > 0 and len(res["buckets"]) == 4 and all(b["allreduce_ms"] > 0 for b in res["buckets"])
  finally:
    dist.destroy_process_group()


@pytest.mark.gpu
def test_overlapped_step_equals_single_stream_step():
  """The executor spreads independent parts of the step over HIP streams of its own (the real half of the perceptual trunk and the
  generator's foreground encoder branch under the generator forward; the discriminator-loss pass and the foreground branch's
  backward under the generator-loss pass; each with its own gradient / scratch buffers): bit-identical to the whole step on one
  stream, over repeated steps (no buffer of one stream may leak into another)."""
  from voicepuppet_amd import _lib
  L = _lib.lib()
  ngf = ndf = 8
  p = make_params(ngf, ndf, 5)
  batch = [torch.tensor(b, device="cuda") for b in synth(2, 256, 13)]
  for dtype in ("f32", "bf16"):
    runs = []
    for overlap in (1, 0):
      try:
        e = PixReferEngine(2, 256, ngf, ndf, dtype=dtype, training=True)
        e.set_option("overlap", overlap)
        e.load_params(p)
        trace = []
        for _ in range(3):
          e.forward(*batch)
          e.backward()
          torch.cuda.synchronize()
          trace.append((e.grads_d.clone(), e.grads_g.clone(), dict(e.losses())))
          e.adam_step(3e-4)
        torch.cuda.synchronize()
        runs.append((trace, e.params_g.clone(), e.params_d.clone()))
      finally:
        pass
    (ta, ga, da), (tb, gb, db) = runs
    for (d1, g1, l1), (d2, g2, l2) in zip(ta, tb):
      assert torch.equal(d1, d2) and torch.equal(g1, g2) and l1 == l2, dtype
    assert torch.equal(ga, gb) and torch.equal(da, db)


@pytest.mark.gpu
def test_overlapped_step_equals_single_stream_step_at_full_width_batch_8():
  """The same at ngf = ndf = 64 and batch 8, where the half-batch launches of the perceptual trunk (8 images) and its full-batch
  launches (16) fall on different sides of the patch kernels' minimum-grid rule: the half-batch plans follow the full-batch kernel
  choice, so the overlapped and the single-stream step still agree bit for bit (round 2: they did not - 2e-4 on the losses)."""
  from voicepuppet_amd import _lib
  L = _lib.lib()
  g = torch.Generator(device="cuda").manual_seed(3)
  batch = [torch.rand(8, 256, 256, c, device="cuda", generator=g) for c in (6, 6, 3, 3)]
  runs = []
  for overlap in (1, 0):
    try:
      e = PixReferEngine(8, 256, 64, 64, dtype="bf16", training=True, streams=0 if overlap else 1)    # (the descriptor form of the switch)
      e.load_params(e.random_params(seed=0))
      for _ in range(2):
        e.train_step(*batch, lr=3e-4)
      torch.cuda.synchronize()
      runs.append((e.grads_g.clone(), e.grads_d.clone(), e.params_g.clone(), e.params_d.clone(), dict(e.losses())))
    finally:
      pass
  a, b = runs
  assert a[4] == b[4]
  assert all(torch.equal(x, y) for x, y in zip(a[:4], b[:4]))


@pytest.mark.gpu
def test_fused_backward_update_equals_separate_calls():
  """vp_pixrefer_backward_update (backward + Adam x 2 + weight re-pack, bucket by bucket under the backward pass) leaves exactly the
  parameters, Adam slots and next-step losses of vp_pixrefer_backward + vp_adam_tf x 2, over repeated steps and both dtypes."""
  ngf = ndf = 8
  p = make_params(ngf, ndf, 7)
  batch = [torch.tensor(b, device="cuda") for b in synth(2, 256, 21)]
  for dtype in ("f32", "bf16"):
    runs = []
    for fused in (True, False):
      e = PixReferEngine(2, 256, ngf, ndf, dtype=dtype, training=True)
      e.fused_update = fused
      e.load_params(p)
      losses = []
      for _ in range(4):
        e.train_step(*batch, lr=3e-4)
        losses.append(dict(e.losses()))
      torch.cuda.synchronize()
      runs.append((losses, e.params_g.clone(), e.params_d.clone(), [t.clone() for t in e.adam["g"] + e.adam["d"]], e.t_g, e.t_d))
    a, b = runs
    assert a[0] == b[0], dtype
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), dtype
    assert all(torch.equal(x, y) for x, y in zip(a[3], b[3])) and a[4:] == b[4:]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,n,tol_stat,tol_grad,min_launches", [("f32", 2, 2e-5, 2e-4, 3), ("bf16", 8, 2e-5, 2e-2, 8)])
def test_batch_norm_backward_sums_from_the_gradient_epilogue_in_situ(dtype, n, tol_stat, tol_grad, min_launches):
  """Round 6: the launch that completes the gradient of a batch-normalised tensor also produces sum dz and sum dz * zhat in its epilogue
  (staged_epilogue STATS == 2; plain, two-output, 2x2-tap and 4x4 patch launches of a full-width plan: generator encoders / decoders, the
  discriminator's three-group pass and its one-group generator-loss pass), so the tensor's batch-norm backward starts at its finalize.
  Against the same plan with the separate reduce pass (vp_pixrefer_set_option "bwd_sums_in_epilogue" 0): dgamma / dbeta of every
  batch-norm (they ARE the sums) to float32 rounding of a different summation order, every gradient downstream to the same plus - bf16 -
  the handful of roundings of dy that a 1e-7 change of its two coefficients flips."""
  grads, counts = {}, {}
  for on in (1, 0):
    eng = PixReferEngine(n, 256, 64, 64, dtype=dtype, training=True)
    eng.load_params(eng.random_params(5))
    eng.set_option("bwd_sums_in_epilogue", 2 * on)         # 2: every launch that can carry them (the default keeps to the classes where it pays)
    g = torch.Generator(device="cpu").manual_seed(9)
    batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
    eng.forward(*batch); eng.backward()
    torch.cuda.synchronize()
    counts[on] = int(eng.L.vp_pixrefer_counter(eng.h, b"bwd_sums_launches"))
    grads[on] = (eng.get_params(1, src=eng.grads_d), eng.get_params(0, src=eng.grads_g))
    del eng
  # generator: encoder_2/3(/4 from 8 frames up), their foreground twins, the wide decoders; discriminator: layer_2, layer_3 (+ layer_4 on the
  # tap path's 1x1 product) in both passes
  assert counts[0] == 0 and counts[1] >= min_launches, counts
  worst = {}
  for which in (0, 1):
    for k, ref_v in grads[0][which].items():
      got = grads[1][which][k]
      if not np.abs(ref_v).max() > 0:
        assert not np.abs(got).max() > 0, k          # (the analytically zero bias gradients in front of a batch-norm)
        continue
      e = gu.rel_l2(got, ref_v)
      kind = "stat" if k.endswith(("gamma", "beta")) else "grad"
      worst[kind] = max(worst.get(kind, 0.0), e)
      assert e < tol_grad, (k, e)
  # the sums themselves, on the tensors whose gradient does not pass through an earlier fused batch-norm: the discriminator's deepest
  # batch-norm (layer_4) and the generator's last (merged2_decoder_2)
  for which, k in ((0, "discriminator/layer_4/batch_normalization/"), (1, "generator/merged2_decoder_2/batch_normalization/")):
    for f in ("gamma", "beta"):
      e = gu.rel_l2(grads[1][which][k + f], grads[0][which][k + f])
      assert e < tol_stat, (k + f, e)
  print("bwd sums in the epilogue (%s): %d launches, worst rel-L2 vs the reduce pass: %s" % (dtype, counts[1], worst))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [24, 8])
def test_one_output_channel_backward_kernel_in_situ(n):
  """Round 6: the backward-data pass of the PatchGAN's last layer (4x4, stride 1, 512 -> 1 channel) runs on its own kernel
  (conv_cout1.hip: one MFMA step per 16 channels, the lrelu' product, and the two sums of layer_4's batch-norm backward from the same
  launch; its weight gradient on cout1_wgrad_kernel: a wave per pixel, no tap spreading) in both passes - the discriminator-loss pass over the three applications and the generator-loss pass over the fake one.  Against
  the same full-width bf16 plan on the generic implicit-GEMM kernel + the reduce pass (vp_tune "cout1_bwd" 0): every discriminator and
  generator gradient to the roundings a different summation order flips (the K = 16 products are exact in float32, the tap sums are not
  associative), layer_4's dgamma / dbeta - the sums themselves - tightly.  n = 24: both passes on the kernel; n = 8: the
  discriminator-loss pass only (the generator-loss pass is below the kernel's 16384-pixel floor); 961 pixels per image: the last 16-pixel
  tile of every batch-norm group is ragged."""
  L = _lib.lib()
  grads = {}
  try:
    for on in (1, 0):
      L.vp_tune(b"cout1_bwd", 256 if on else 0)
      eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
      eng.load_params(eng.random_params(6))
      g = torch.Generator(device="cpu").manual_seed(10)
      batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
      eng.profile(1)
      eng.forward(*batch); eng.backward()
      torch.cuda.synchronize()
      classes = {r["name"] for r in eng.profile_collect()}
      eng.profile(0)
      assert any(c.startswith("cout1bwd_") for c in classes) == bool(on), classes
      assert any(c.startswith("cout1wgrad_") for c in classes) == bool(on), classes
      grads[on] = (eng.get_params(1, src=eng.grads_d), eng.get_params(0, src=eng.grads_g))
      del eng
  finally:
    L.vp_tune(b"cout1_bwd", -1)
  worst = 0.0
  for which in (0, 1):
    for k, ref_v in grads[0][which].items():
      got = grads[1][which][k]
      if not np.abs(ref_v).max() > 0:
        assert not np.abs(got).max() > 0, k
        continue
      e = gu.rel_l2(got, ref_v)
      worst = max(worst, e)
      assert e < 2e-2, (k, e)
  for f in ("gamma", "beta"):
    k = "discriminator/layer_4/batch_normalization/" + f
    e = gu.rel_l2(grads[1][0][k], grads[0][0][k])
    assert e < 1e-3, (k, e)
  print("one-output-channel backward kernel, n = %d: worst gradient rel-L2 against the generic kernels %.2e" % (n, worst))


@pytest.mark.gpu
def test_engine_close_releases_the_plan_and_later_calls_fail_loudly():
  """PixReferEngine.close() destroys the plan (and its HIP streams) at once - a process that builds several engines closes the ones it
  is done with (EXPERIMENTS.md 0.8 of round 6: an engine created beside the live streams of an earlier one runs slow).  Closing twice is
  harmless; a step on a closed engine is an error from the library, not a crash; a new engine of the same shape reproduces the first
  one's step bit for bit."""
  p = ref.init_params(8, 8, seed=2, dtype=np.float32)
  rng = np.random.default_rng(4)
  batch = [torch.tensor(rng.uniform(size=(1, 256, 256, c)).astype(np.float32), device="cuda") for c in (6, 6, 3, 3)]
  out = []
  for _ in range(2):
    eng = PixReferEngine(1, 256, 8, 8, dtype="f32", training=True)
    eng.load_params(p)
    eng.train_step(*batch, lr=3e-4)
    torch.cuda.synchronize()
    out.append((eng.params_g.clone(), eng.params_d.clone(), dict(eng.losses())))
    eng.close()
    eng.close()
    with pytest.raises(RuntimeError):
      eng.train_step(*batch, lr=3e-4)
  assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


@pytest.mark.gpu
def test_executor_streams_are_process_wide_and_shared_by_every_plan():
  """Round 6: the executor's three extra HIP streams are created once per process (vp_reserve_streams, or the first training plan) and
  shared by every plan - streams created behind an RCCL communicator's or an earlier plan's got the runtime's leftover hardware queues
  and ran the step 8 - 30 % slow (scripts/exp_dp_order.py, scripts/exp_engine_sequence.py).  Two engines of one process report the same
  side stream, before and after one of them is closed; reserving again changes nothing; two engines stepping alternately on the shared
  streams stay bit-identical to an engine stepping alone."""
  L = _lib.lib()
  _lib.check(L.vp_reserve_streams())
  p = ref.init_params(8, 8, seed=3, dtype=np.float32)
  rng = np.random.default_rng(5)
  batch = [torch.tensor(rng.uniform(size=(1, 256, 256, c)).astype(np.float32), device="cuda") for c in (6, 6, 3, 3)]

  def make():
    e = PixReferEngine(1, 256, 8, 8, dtype="bf16", training=True)
    e.load_params(p)
    return e
  a, b = make(), make()
  sa, sb = int(L.vp_pixrefer_side_stream(a.h)), int(L.vp_pixrefer_side_stream(b.h))
  assert sa == sb and sa != 0
  for _ in range(3):
    a.train_step(*batch, lr=3e-4); b.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  assert torch.equal(a.params_g, b.params_g) and torch.equal(a.params_d, b.params_d)
  got = (a.params_g.clone(), a.params_d.clone())
  a.close()
  _lib.check(L.vp_reserve_streams())
  c = make()
  assert int(L.vp_pixrefer_side_stream(c.h)) == sb
  for _ in range(3):
    c.train_step(*batch, lr=3e-4)
  torch.cuda.synchronize()
  assert torch.equal(c.params_g, got[0]) and torch.equal(c.params_d, got[1])
  b.close(); c.close()
