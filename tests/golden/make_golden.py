#!/usr/bin/env python
"""Generates the committed fixtures under tests/golden/ (run in the BUILD container only):

  sample22_256.npz   the reference's inference fixture sample/22.jpg (1536x512 triptych: frame | 3-D face | matte)
                     and background/1.jpg, decoded HERE with PIL and resized to 256x256 uint8, so that no JPEG
                     decoder has to agree on the GPU box (BASELINE config 1 input)
  ops.npz            toy-size conv / deconv / batch-norm / pool forward+backward results of the float64 oracle
  mini_step.npz      one G+D step of a mini PixReferNet (ngf=ndf=8, N=1, 256x256) on sample22: losses, output
                     crop, per-tensor gradient norms, post-Adam parameter checksums
  full_width.npz     config 1 proper: the generator forward on sample22 at ngf = 64; and three consecutive full G+D steps at
                     ngf = ndf = 64, N = 1 (losses per step, step-1 output, per-tensor gradient norms, post-step parameter sums)
                     (`python tests/golden/make_golden.py full`: ~10 minutes)
  full_width_n4.npz  the same three steps on a batch of FOUR different samples (every batch-norm has real statistics, the 1x1
                     bottleneck included), with a strided sample of every step-1 gradient tensor
                     (`python tests/golden/make_golden.py full_n4`: ~10 minutes)
  full_width_512_n2.npz  ONE step at 512 x 512 (BASELINE config 4's image size) on two samples, same contents as full_width_n4
                     (`python tests/golden/make_golden.py full_512`: ~10 minutes)
  frame_pack.npz     input pipeline: uint8 triptych frames + crops -> packed float tensors by the host path (PIL bilinear)
  logmel.npz         log-mel of a seeded 4096-sample chirp+noise, and the 257x80 mel matrix
  bfmnet.npz         BFMNet coefficients for a seeded 5-frame clip (parameters regenerated from the seed)
  bfm_recon.npz      outputs of the REFERENCE's own utils/reconstruct_mesh.py (pure numpy, imported from /root/reference here)
                     for a seeded synthetic face model and a 5-frame clip, the float32 / integer packing of
                     infer_bfmvid.py:92-99, and the frames the compiled reference rasteriser drew from them: reference-captured
  raster.npz         a synthetic mesh (oracle.raster_ref.synthetic_mesh) and the image / mask / depth the REFERENCE's own
                     compiled mesh_core.cpp (oracle/_ref, built by oracle/Makefile) rasterised from it: reference-captured

The reference itself cannot produce vectors for this path (TF1.x is not installable: SURVEY.md 8c), so these
pin the ORACLE against drift and give the GPU tests fixed inputs; they are not reference-captured outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import audio_ref as ar
from oracle import nn_ops as ops
from oracle import pixrefer_ref as ref

REF = "/root/reference"


def sample22():
  from PIL import Image
  img = Image.open(os.path.join(REF, "sample", "22.jpg")).convert("RGB")
  assert img.size == (1536, 512)
  panels = [np.asarray(img.crop((k * 512, 0, (k + 1) * 512, 512)).resize((256, 256), Image.BILINEAR)) for k in range(3)]
  bg = np.asarray(Image.open(os.path.join(REF, "background", "1.jpg")).convert("RGB").resize((256, 256), Image.BILINEAR))
  np.savez_compressed(os.path.join(HERE, "sample22_256.npz"), frame=panels[0], face3d=panels[1], matte=panels[2], background=bg)
  return panels, bg


def toy_ops():
  rng = np.random.default_rng(42)
  d = {}
  x, w, b = rng.normal(size=(2, 8, 8, 4)), rng.normal(size=(4, 4, 4, 6)), rng.normal(size=6)
  d["conv_x"], d["conv_w"], d["conv_b"] = x, w, b
  d["conv_s2_y"] = ops.conv2d_fwd(x, w, b, 2, 1)
  dy = rng.normal(size=d["conv_s2_y"].shape)
  d["conv_s2_dy"] = dy
  d["conv_s2_dx"], d["conv_s2_dw"], d["conv_s2_db"] = ops.conv2d_bwd(x, w, dy, 2, 1)
  d["conv_s1_y"] = ops.conv2d_fwd(x, w, b, 1, 1)
  wd = rng.normal(size=(4, 4, 5, 4))
  d["deconv_w"] = wd
  d["deconv_y"] = ops.deconv4s2_fwd(x, wd, None)
  dyd = rng.normal(size=d["deconv_y"].shape)
  d["deconv_dy"] = dyd
  d["deconv_dx"], d["deconv_dw"], _ = ops.deconv4s2_bwd(x, wd, dyd)
  g, bt = rng.normal(1, 0.1, 4), rng.normal(0, 0.1, 4)
  z, cache = ops.bn_train_fwd(x, g, bt)
  d["bn_gamma"], d["bn_beta"], d["bn_z"] = g, bt, z
  dz = rng.normal(size=x.shape)
  d["bn_dz"] = dz
  d["bn_dy"], d["bn_dgamma"], d["bn_dbeta"] = ops.bn_train_bwd(dz, cache)
  y, idx = ops.maxpool2x2_fwd(x)
  d["pool_y"] = y
  d["pool_dx"] = ops.maxpool2x2_bwd(y * 0 + 1.5, idx, x.shape)
  np.savez_compressed(os.path.join(HERE, "ops.npz"), **d)


def mini_step(panels, bg):
  ngf = ndf = 8
  frame, face3d, matte = [p.astype(np.float64) / 255.0 for p in panels]
  # inference-style packing of infer_bfmvid.py:175-178,226-229 used as a training sample (example == current)
  inputs = np.concatenate([face3d, face3d], axis=-1)[None]
  fg = np.concatenate([frame * matte, frame * matte], axis=-1)[None]
  targets, masks = frame[None], matte[None]
  p = ref.init_params(ngf, ndf, seed=7)
  st = ref.TrainState({k: v.copy() for k, v in p.items()}, ngf, ndf)
  nodes = st.step(inputs, fg, targets, masks)
  inf = ref.inference(p, inputs, fg[..., :3], bg[None].astype(np.float64) / 255.0, ngf)
  d = {"seed": 7, "ngf": ngf}
  for k in ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss"):
    d[k] = np.float64(nodes[k])
  d["Outputs_crop"] = nodes["Outputs"][0, 96:160, 96:160]
  d["Infer_Outputs_crop"] = inf["Outputs"][0, 96:160, 96:160]
  d["Infer_Outputs_mean"] = inf["Outputs"].mean(axis=(0, 1, 2))
  names = sorted(nodes["Gen_grads"]) + sorted(nodes["Discrim_grads"])
  d["grad_names"] = np.array(names)
  d["grad_norms"] = np.array([np.linalg.norm(nodes["Gen_grads" if n.startswith("generator") else "Discrim_grads"][n]) for n in names])
  d["param_sums_after"] = np.array([st.p[n].sum() for n in names])
  np.savez_compressed(os.path.join(HERE, "mini_step.npz"), **d)


def full_width(panels, bg):
  """BASELINE config 1 proper (generator forward on sample/22.jpg at ngf = 64) and THREE consecutive full G+D steps at the
  benchmark width (ngf = ndf = 64, N = 1, 256x256) from the float64 oracle.  The parameters are float32 draws (what the device
  holds) promoted to float64, regenerated from the seed by the tests; ~10 minutes of numpy on 8 cores."""
  ngf = ndf = 64
  seed = 9
  frame, face3d, matte = [p.astype(np.float64) / 255.0 for p in panels]
  inputs = np.concatenate([face3d, face3d], axis=-1)[None]
  fg = np.concatenate([frame * matte, frame * matte], axis=-1)[None]
  targets, masks = frame[None], matte[None]
  p = {k: v.astype(np.float64) for k, v in ref.init_params(ngf, ndf, seed=seed, dtype=np.float32).items()}
  inf = ref.inference(p, inputs, fg[..., :3], bg[None].astype(np.float64) / 255.0, ngf)
  d = {"seed": seed, "ngf": ngf, "Infer_Outputs": inf["Outputs"][0].astype(np.float32), "Infer_Alphas_mean": np.float64(inf["Alphas"].mean())}
  # f32_probs: the GAN terms as the reference's float32 graph evaluates them (the discriminator saturates after one step here:
  # see oracle.pixrefer_ref.forward_backward); everything else in float64
  st = ref.TrainState({k: v.copy() for k, v in p.items()}, ngf, ndf, f32_probs=True)
  keys = ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss")
  sat = []
  losses, sums, norms2 = [], [], []
  names = None
  for step in range(3):
    nodes = st.step(inputs, fg, targets, masks)
    if names is None:
      names = sorted(nodes["Gen_grads"]) + sorted(nodes["Discrim_grads"])
      d["grad_names"] = np.array(names)
      d["Outputs"] = nodes["Outputs"][0].astype(np.float32)
      d["grad_norms"] = np.array([np.linalg.norm(nodes["Gen_grads" if n.startswith("generator") else "Discrim_grads"][n]) for n in names])
    losses.append([nodes[k] for k in keys])
    sat.append(int((nodes["Predict_fake"] == 1).sum()))
    sums.append([st.p[n].sum() for n in names])
    norms2.append([np.linalg.norm(st.p[n] - p[n]) for n in names])     # size of the update since the start
    print("full-width step", step, dict(zip(keys, losses[-1])), flush=True)
  d["losses"] = np.array(losses)
  d["saturated_fake_predictions"] = np.array(sat)
  d["param_sums_after"] = np.array(sums)
  d["update_norms_after"] = np.array(norms2)
  np.savez_compressed(os.path.join(HERE, "full_width.npz"), **d)


def grad_sample_index(size, want):
  """Deterministic strided subsample of a flattened gradient tensor (the full-width tensors hold up to 8.4 M floats each)."""
  stride = max(1, size // want)
  return np.arange(0, size, stride)[:want]


BOTTLENECK = ("merged_encoder_2", "merged_encoder_3", "merged_encoder_4", "merged_encoder_5",
              "merged_decoder_5", "merged_decoder_4", "merged_decoder_3", "merged_decoder_2")


def full_width_batch(panels, bg, nsamp=4):
  """the N = 1 fixture leaves the two deepest layers with zero-variance batch statistics and exactly-zero weight
  gradients (and N = 2 is hardly better: a batch-norm over two values has an analytically vanishing backward pass, so the encoder
  gradients below it are rounding noise).  This one runs THREE consecutive G+D steps at ngf = ndf = 64 on a batch of FOUR different
  samples (sample/22.jpg and three mirrored / shifted / re-lit variants), so every batch-norm has real statistics and every layer
  a real gradient; besides losses, output crops, gradient norms and post-step parameter sums / update norms it keeps a strided
  SAMPLE of every step-1 gradient tensor (32768 values of the eight bottleneck kernels, 4096 of the others) so the device can be
  compared element-wise, not only by norm.  The batch is also the 4-per-GPU share of the 8-GPU strong-scaling run."""
  ngf = ndf = 64
  seed = 13
  frame, face3d, matte = [p.astype(np.float64) / 255.0 for p in panels]
  # four different samples: the original, a mirrored re-lit one, a shifted darker one, an upside-down one with a shrunken matte
  variants = [(frame, face3d, matte),
              (frame[:, ::-1] * 0.8 + 0.1, np.roll(face3d[:, ::-1], 7, axis=0), matte[:, ::-1]),
              (np.roll(frame, 19, axis=1) * 0.6, np.roll(face3d, 19, axis=1) ** 1.5, np.roll(matte, 19, axis=1)),
              (frame[::-1] * 0.5 + 0.4, face3d[::-1], matte[::-1] * (np.roll(matte[::-1], 11, axis=0) > 0.5))][:nsamp]
  inputs = np.stack([np.concatenate([face3d, v[1]], axis=-1) for v in variants])
  fg = np.stack([np.concatenate([frame * matte, v[0] * v[2]], axis=-1) for v in variants])
  targets, masks = np.stack([v[0] for v in variants]), np.stack([v[2] for v in variants])
  p = {k: v.astype(np.float64) for k, v in ref.init_params(ngf, ndf, seed=seed, dtype=np.float32).items()}
  d = {"seed": seed, "ngf": ngf, "inputs": (inputs * 255).round().astype(np.uint8), "fg_inputs": (fg * 255).round().astype(np.uint8),
       "targets": (targets * 255).round().astype(np.uint8), "masks": (masks * 255).round().astype(np.uint8)}
  # the device is fed exactly these uint8 / 255 values
  inputs, fg, targets, masks = [d[k].astype(np.float64) / 255.0 for k in ("inputs", "fg_inputs", "targets", "masks")]
  st = ref.TrainState({k: v.copy() for k, v in p.items()}, ngf, ndf, f32_probs=True)
  keys = ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss")
  losses, sums, norms2, sat = [], [], [], []
  names = None
  for step in range(3):
    nodes = st.step(inputs, fg, targets, masks)
    if names is None:
      names = sorted(nodes["Gen_grads"]) + sorted(nodes["Discrim_grads"])
      d["grad_names"] = np.array(names)
      d["Outputs_crop"] = nodes["Outputs"][:, 64:192, 64:192].astype(np.float32)
      grads = [nodes["Gen_grads" if n.startswith("generator") else "Discrim_grads"][n] for n in names]
      d["grad_norms"] = np.array([np.linalg.norm(g) for g in grads])
      samp = []
      for n, g in zip(names, grads):
        want = 32768 if (n.endswith("kernel") and any(("/%s/" % b) in n for b in BOTTLENECK)) else 4096
        samp.append(g.reshape(-1)[grad_sample_index(g.size, want)])
      d["grad_sample_sizes"] = np.array([len(x) for x in samp])
      d["grad_samples"] = np.concatenate(samp).astype(np.float32)
    losses.append([nodes[k] for k in keys])
    sat.append(int((nodes["Predict_fake"] == 1).sum()))
    sums.append([st.p[n].sum() for n in names])
    norms2.append([np.linalg.norm(st.p[n] - p[n]) for n in names])
    print("full-width N=%d step" % nsamp, step, dict(zip(keys, losses[-1])), flush=True)
  d["losses"] = np.array(losses)
  d["saturated_fake_predictions"] = np.array(sat)
  d["param_sums_after"] = np.array(sums)
  d["update_norms_after"] = np.array(norms2)
  np.savez_compressed(os.path.join(HERE, "full_width_n%d.npz" % nsamp), **d)


def batch_512(frame, face3d, matte):
  """BASELINE config 4's image size at a batch the float64 oracle can afford: TWO 512 x 512 samples (sample/22.jpg at its native size
  and a mirrored, re-lit variant), quantised to uint8 exactly as the device is fed.  (At 512 x 512 the deepest tensor is 2 x 2, so a
  batch of two still gives every batch-norm eight or more values per channel - unlike N = 2 at 256 x 256.)
  tests/test_gpu_fullwidth.py rebuilds the batch from the three stored panels with the same four lines."""
  f, a, m = [x.astype(np.float64) / 255.0 for x in (frame, face3d, matte)]
  variants = [(f, a, m), (f[:, ::-1] * 0.8 + 0.1, np.roll(a[:, ::-1], 7, axis=0), m[:, ::-1])]
  inputs = np.stack([np.concatenate([a, v[1]], axis=-1) for v in variants])
  fg = np.stack([np.concatenate([f * m, v[0] * v[2]], axis=-1) for v in variants])
  targets, masks = np.stack([v[0] for v in variants]), np.stack([v[2] for v in variants])
  return [(x * 255).round().astype(np.uint8) for x in (inputs, fg, targets, masks)]


def full_width_512():
  """config 4 (512 x 512, 8 per GPU) was checked through size-independent properties only.  ONE G+D step at ngf = ndf = 64
  on the two-sample 512 x 512 batch above: losses, output crops, per-tensor gradient norms and the strided gradient samples of
  full_width_batch.  (`python tests/golden/make_golden.py full_512`: ~10 minutes.)"""
  from PIL import Image
  img = Image.open(os.path.join(REF, "sample", "22.jpg")).convert("RGB")
  assert img.size == (1536, 512)
  frame, face3d, matte = [np.asarray(img.crop((k * 512, 0, (k + 1) * 512, 512))) for k in range(3)]
  ngf = ndf = 64
  seed = 17
  u8 = batch_512(frame, face3d, matte)
  inputs, fg, targets, masks = [x.astype(np.float64) / 255.0 for x in u8]
  p = {k: v.astype(np.float64) for k, v in ref.init_params(ngf, ndf, seed=seed, dtype=np.float32).items()}
  st = ref.TrainState({k: v.copy() for k, v in p.items()}, ngf, ndf, f32_probs=True)
  nodes = st.step(inputs, fg, targets, masks)
  keys = ("Discrim_loss", "Gen_loss_GAN", "Gen_loss_L1", "Gen_loss", "Perceptual_loss")
  names = sorted(nodes["Gen_grads"]) + sorted(nodes["Discrim_grads"])
  grads = [nodes["Gen_grads" if n.startswith("generator") else "Discrim_grads"][n] for n in names]
  samp = []
  for n, g in zip(names, grads):
    want = 32768 if (n.endswith("kernel") and any(("/%s/" % b) in n for b in BOTTLENECK)) else 4096
    samp.append(g.reshape(-1)[grad_sample_index(g.size, want)])
  d = {"seed": seed, "ngf": ngf, "frame": frame, "face3d": face3d, "matte": matte, "grad_names": np.array(names),
       "losses": np.array([nodes[k] for k in keys]), "Outputs_crop": nodes["Outputs"][:, 192:320, 192:320].astype(np.float32),
       "grad_norms": np.array([np.linalg.norm(g) for g in grads]), "grad_sample_sizes": np.array([len(x) for x in samp]),
       "grad_samples": np.concatenate(samp).astype(np.float32), "saturated_fake_predictions": int((nodes["Predict_fake"] == 1).sum())}
  print("full-width 512x512 N=2", dict(zip(keys, d["losses"])), flush=True)
  np.savez_compressed(os.path.join(HERE, "full_width_512_n2.npz"), **d)


def frame_pack():
  """Input-pipeline fixture: uint8 triptych frames + crops -> the four packed float tensors, computed by the
  host path (PIL bilinear on float planes standing in for cv2.resize, generator.py:956-1019)."""
  from oracle.input_pack_ref import pack_frames_ref as host_pack_reference
  S, N = 32, 4
  rng = np.random.default_rng(31)
  yy, xx = np.mgrid[0:S, 0:3 * S]
  ex = np.stack([np.clip(rng.integers(0, 256, (S, 3 * S, 3)) * 0.5 + 64 * np.sin(yy / 3.0 + k)[..., None] + 64, 0, 255) for k in range(N)]).astype(np.uint8)
  cur = rng.integers(0, 256, (N, S, 3 * S, 3)).astype(np.uint8)
  crops = np.array([[[0, 0, S], [0, 0, S]],                       # no crop: the resize is the identity
                    [[0, 3, 29], [3, 0, 29]],                     # int(0.9 * 32) = 28 .. 32
                    [[4, 4, 28], [0, 0, 28]],
                    [[1, 2, 30], [1, 0, 31]]], np.int32)
  assert (crops[..., 0] + crops[..., 2] <= S).all() and (crops[..., 1] + crops[..., 2] <= S).all()
  outs = [host_pack_reference(ex[i], cur[i], crops[i], S) for i in range(N)]
  np.savez_compressed(os.path.join(HERE, "frame_pack.npz"), ex=ex, cur=cur, crops=crops,
                      inputs=np.stack([o[0] for o in outs]).astype(np.float32), fg_inputs=np.stack([o[1] for o in outs]).astype(np.float32),
                      targets=np.stack([o[2] for o in outs]).astype(np.float32), masks=np.stack([o[3] for o in outs]).astype(np.float32))


def audio():
  rng = np.random.default_rng(11)
  t = np.arange(4096) / 16000.0
  pcm = np.clip(0.1 * rng.normal(size=(1, 4096)) + 0.3 * np.sin(2 * np.pi * (100 + 3900 * t / t[-1] / 2) * t), -1, 1)
  np.savez_compressed(os.path.join(HERE, "logmel.npz"), pcm=pcm.astype(np.float32), logmel=ar.extract_mfcc(pcm.astype(np.float32).astype(np.float64)),
                      mel_matrix=ar.linear_to_mel_weight_matrix())
  p = ar.init_bfmnet_params(21)
  pcm5 = np.clip(0.2 * rng.normal(size=(1, ar.pcm_length_for(5))), -1, 1).astype(np.float32)
  mf = ar.extract_mfcc(pcm5.astype(np.float64))
  ears = np.full((1, 5, 1), 0.003)
  out = ar.bfmnet_fwd(p, ears, mf, [5])
  np.savez_compressed(os.path.join(HERE, "bfmnet.npz"), seed=21, pcm=pcm5, ears=ears, mfcc=mf, coeff=out["BFMCoeffDecoder"], enc=out["MfccEncoder"])


def raster():
  from oracle import raster_ref as rr
  assert rr.have_compiled_reference(), "run `make -C oracle` first"
  d = {}
  for tag, (seed, nlat, nlon, h, w) in {"a": (3, 40, 60, 224, 224), "b": (5, 12, 16, 96, 128)}.items():
    v, t, c = rr.synthetic_mesh(seed, nlat, nlon, h, w)
    img, mask, depth = rr.render_colors_ref(v, t, c, h, w)
    d.update({tag + "_vertices": v, tag + "_triangles": t, tag + "_colors": c, tag + "_image": img, tag + "_mask": mask, tag + "_depth": depth})
  np.savez_compressed(os.path.join(HERE, "raster.npz"), **d)


def bfm_recon():
  from oracle import bfm_ref as br
  from oracle import raster_ref as rr
  sys.path.insert(0, os.path.join(REF, "utils"))
  import reconstruct_mesh as rm                        # the reference module itself (numpy only)
  assert rr.have_compiled_reference(), "run `make -C oracle` first"
  d = {"model_seed": 3, "coeff_seed": 5}
  fm = br.synthetic_facemodel(3)
  coeff, angles = br.synthetic_coeffs(5, 5)
  d["coeff"], d["angles"] = coeff, angles
  d["model_checksum"] = np.array([fm.idBase.sum(), fm.exBase.sum(), fm.texBase.sum(), fm.meanshape.sum(), fm.meantex.sum(),
                                  float(fm.tri.sum()), float(fm.point_buf.sum()), float(fm.keypoints.sum())])
  names = ["face_shape", "face_texture", "face_color", "face_projection", "z_buffer", "landmarks_2d"]
  outs = {n: [] for n in names}
  verts, cols, imgs, masks = [], [], [], []
  for t in range(coeff.shape[0]):
    res = rm.Reconstruction_rotation(coeff[t:t + 1], fm, angles[t:t + 1])
    for n, r in zip(names, res):
      outs[n].append(r[0])
    # infer_bfmvid.py:92-108
    shape = np.squeeze(np.concatenate([res[3], res[4]], axis=2), (0))
    color = np.clip(np.squeeze(res[2], (0)), 0, 255).astype(np.int32)
    v, c = shape.reshape(-1).astype(np.float32).copy(), color.reshape(-1).astype(np.float32).copy()
    img, mask, _ = rr.render_colors_ref(v, (fm.tri - 1).reshape(-1).astype(np.int32), c, 224, 224)
    verts.append(v.reshape(-1, 3)); cols.append(c.reshape(-1, 3)); imgs.append(img); masks.append(mask)
  for n in names:
    d[n] = np.stack(outs[n])
  d["vertices"], d["colors"], d["images"], d["masks"] = np.stack(verts), np.stack(cols), np.stack(imgs), np.stack(masks)
  np.savez_compressed(os.path.join(HERE, "bfm_recon.npz"), **d)


if __name__ == "__main__":
  if sys.argv[1:] == ["frame_pack"]:
    frame_pack()
    sys.exit(0)
  if sys.argv[1:] == ["full"]:
    full_width(*sample22())
    sys.exit(0)
  if sys.argv[1:] == ["full_n4"]:
    full_width_batch(*sample22(), nsamp=4)
    sys.exit(0)
  if sys.argv[1:] == ["full_512"]:
    full_width_512()
    sys.exit(0)
  if sys.argv[1:] == ["raster"]:
    raster()
    bfm_recon()
    sys.exit(0)
  panels, bg = sample22()
  toy_ops()
  frame_pack()
  mini_step(panels, bg)
  audio()
  raster()
  bfm_recon()
  for f in sorted(os.listdir(HERE)):
    print(f, os.path.getsize(os.path.join(HERE, f)))
