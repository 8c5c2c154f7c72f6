"""-m gpu: the reference-shaped Python surface (PixReferNet / BFMNet / DataGenerator / Session, the two CLI
entry points) driving the HIP executors."""
import os

import numpy as np
import pytest
import torch

import gpu_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config", "params.yml")


def test_train_cli_runs_and_checkpoints(tmp_path, monkeypatch, capsys):
  from voicepuppet_amd.pixrefer import train_pixrefer
  monkeypatch.chdir(tmp_path)
  os.makedirs("config")
  train_pixrefer.main(["--config_path", CFG, "--steps", "3", "--batch_size", "1", "--img_size", "256"])
  assert os.path.isdir("ckpt_pixrefer") and os.path.isdir("log/summary_pixrefer")


def test_build_train_op_nodes_and_lr_steps():
  from voicepuppet_amd.generator.generator import PixReferDataGenerator
  from voicepuppet_amd.pixrefer.pixrefer import PixReferNet, TRAIN_KEYS
  from voicepuppet_amd.runtime import Session
  gen = PixReferDataGenerator(CFG)
  p = gen.params
  p.batch_size, p.img_size = 2, 256
  it = gen.get_dataset().make_one_shot_iterator()
  net = PixReferNet(CFG)
  p = net.params
  p.batch_size = 2
  p.add_hparam("is_training", True)
  p.sess = Session()
  p.vgg_model_path = "allmodels/vgg_16.ckpt"
  p.amd = dict(p.amd, dtype="f32")
  net.set_params(p)
  nodes = net.build_train_op(*it.get_next())
  assert sorted(nodes) == sorted(TRAIN_KEYS)
  sess = p.sess
  losses = []
  for i in range(3):
    _, g, l1, d, lr, gs = sess.run([nodes["Train_op"], nodes["Gen_loss_GAN"], nodes["Gen_loss_L1"], nodes["Discrim_loss"],
                                    nodes["Lr"], nodes["Global_step"]])
    assert gs == 2 * (i + 1) and lr == pytest.approx(3e-4)
    assert np.isfinite([g, l1, d]).all()
    losses.append(l1)
  out, al, pr = sess.run([nodes["Outputs"], nodes["Alphas"], nodes["Predict_real"]])
  assert out.shape == (2, 256, 256, 3) and al.shape == (2, 256, 256, 3) and pr.shape == (2, 30, 30, 1)
  assert 0 <= al.min() and al.max() <= 1 and 0 < pr.min() and pr.max() < 1
  path = net.save("/tmp/vp_test_ckpt.npz")
  z = np.load(path)
  assert "generator/encoder_1/conv2d/kernel" in z.files and "discriminator/layer_4/conv2d/kernel/Adam_1" in z.files
  assert int(z["global_step"]) == 6


def test_batched_inference_equals_batch_one_loop():
  """per-sample batch-norm statistics: N frames in one launch == the reference's N separate batch-1 runs."""
  from voicepuppet_amd.engine import PixReferEngine
  from oracle import pixrefer_ref as ref
  p = ref.init_params(8, 8, seed=5, dtype=np.float32)
  rng = np.random.default_rng(6)
  x = [torch.tensor(rng.uniform(size=(3, 256, 256, c)).astype(np.float32), device="cuda") for c in (6, 3, 3)]
  e3 = PixReferEngine(3, 256, 8, 8, dtype="f32", training=False, per_sample_bn=True)
  e3.load_params(p)
  e3.forward(*x)
  got = e3.tensor("Outputs_raw").clone()
  e1 = PixReferEngine(1, 256, 8, 8, dtype="f32", training=False)
  e1.load_params(p)
  for i in range(3):
    e1.forward(*[t[i:i + 1].contiguous() for t in x])
    assert gu.rel_l2(got[i].cpu().numpy(), e1.tensor("Outputs_raw")[0].cpu().numpy()) < 1e-5
  out = ref.inference({k: v.astype(np.float64) for k, v in p.items()}, x[0][:1].cpu().numpy().astype(np.float64),
                      x[1][:1].cpu().numpy().astype(np.float64), x[2][:1].cpu().numpy().astype(np.float64), ngf=8)
  assert gu.rel_l2((got[0].cpu().numpy() + 1) / 2, out["Outputs"][0]) < 1e-3


def test_infer_bfmvid_cli_end_to_end(tmp_path, monkeypatch):
  from PIL import Image
  from scipy.io import wavfile
  from voicepuppet_amd.pixrefer import infer_bfmvid
  monkeypatch.chdir(tmp_path)
  rng = np.random.default_rng(0)
  Image.fromarray((rng.uniform(size=(512, 1536, 3)) * 255).astype(np.uint8)).save("face.jpg")
  t = np.arange(8000) / 16000.0
  wavfile.write("a.wav", 16000, (0.3 * np.sin(2 * np.pi * 440 * t) * 32767).astype(np.int16))
  infer_bfmvid.main(["--config_path", CFG, "--frame_batch", "4", "face.jpg", "a.wav"])
  frames = sorted(os.listdir("output"))
  assert len(frames) == int(1 + 8000 / 640)            # pad_len video frames (infer_bfmvid.py:162)
  assert Image.open(os.path.join("output", "0.jpg")).size == (512, 512)


def test_infer_bfmvid_cli_through_the_clip_renderer(tmp_path, monkeypatch):
  """The CLI's ClipRenderer branch end to end (infer_bfmvid.py:79-122,221-243): BFM/BFM_model_front.mat (a synthetic face model
  written with scipy.io.savemat in the layout utils/bfm_load_data.py:9-21 reads) + the photo's coefficient file -> wav -> log-mel ->
  BFMNet -> spliced coefficients -> device reconstruction + rasteriser -> resize / paste -> generator -> frames.  The conditioning
  channels must differ from the fallback run (reference 3-D face panel) and between frames."""
  from PIL import Image
  from scipy.io import savemat, wavfile
  from oracle import bfm_ref as br
  from voicepuppet_amd.pixrefer import infer_bfmvid
  monkeypatch.chdir(tmp_path)
  rng = np.random.default_rng(0)
  Image.fromarray((rng.uniform(size=(512, 1536, 3)) * 255).astype(np.uint8)).save("face.jpg")
  t = np.arange(8000) / 16000.0
  wavfile.write("a.wav", 16000, (0.3 * np.sin(2 * np.pi * 440 * t) * 32767).astype(np.int16))
  fm = br.synthetic_facemodel(3)
  os.makedirs("BFM")
  savemat(os.path.join("BFM", "BFM_model_front.mat"),
          {"meanshape": fm.meanshape, "idBase": fm.idBase, "exBase": fm.exBase, "meantex": fm.meantex, "texBase": fm.texBase,
           "point_buf": fm.point_buf, "tri": fm.tri, "keypoints": (fm.keypoints + 1).reshape(1, -1)})
  coeff, _ = br.synthetic_coeffs(1, 5)
  np.savez("photo.npz", bfmcoeff=coeff.reshape(1, 257), transform_params=np.array([512, 512, 1.0, 0.0, 0.0], np.float32),
           center_x=256, center_y=256, ratio=0.9)
  captured = []
  real = infer_bfmvid.render_faces
  monkeypatch.setattr(infer_bfmvid, "render_faces", lambda *a, **k: captured.append(real(*a, **k)) or captured[-1])
  infer_bfmvid.main(["--config_path", CFG, "--frame_batch", "4", "--bfmcoeff", "photo.npz", "face.jpg", "a.wav"])
  frames = sorted(os.listdir("output"))
  nf = int(1 + 8000 / 640)
  assert len(frames) == nf and Image.open(os.path.join("output", "0.jpg")).size == (512, 512)
  # (the launcher keeps the rendered frames on the device: a uint8 torch tensor)
  assert len(captured) == 1 and tuple(captured[0].shape) == (nf, 512, 512, 3) and captured[0].dtype == torch.uint8 and captured[0].is_cuda
  drawn = captured[0].cpu().numpy().reshape(nf, -1)
  assert (drawn.max(axis=1) > 0).all()                                # every frame has a rasterised face pasted in
  assert any(not np.array_equal(drawn[0], drawn[i]) for i in range(1, nf))   # and the mouth / pose moves over the clip
  with_render = np.asarray(Image.open(os.path.join("output", "1.jpg"))).astype(np.int32)
  infer_bfmvid.main(["--config_path", CFG, "--frame_batch", "4", "--output_dir", "out_fallback", "face.jpg", "a.wav"])
  fallback = np.asarray(Image.open(os.path.join("out_fallback", "1.jpg"))).astype(np.int32)
  assert np.abs(with_render - fallback).max() > 0                     # the generator really was conditioned on the rendered face


def _train_net(batch=1, dtype="f32"):
  from voicepuppet_amd.pixrefer.pixrefer import PixReferNet
  from voicepuppet_amd.runtime import Placeholder, Session
  net = PixReferNet(CFG)
  p = net.params
  p.batch_size = batch
  p.ngf = p.ndf = 8
  p.add_hparam("is_training", True)
  p.sess = Session()
  p.vgg_model_path = "allmodels/vgg_16.ckpt"
  p.amd = dict(p.amd, dtype=dtype)
  net.set_params(p)
  ph = [Placeholder([batch, 256, 256, c], n) for c, n in ((6, "inputs"), (6, "fg"), (3, "targets"), (3, "masks"))]
  nodes = net.build_train_op(*ph)
  return net, p.sess, nodes, ph


@pytest.mark.parametrize("fmt", ["npz", "tf"])
def test_save_restore_resumes_bit_for_bit(tmp_path, fmt):
  """restore() brings back everything save() wrote (parameters, Adam m / v, update counters, global_step):
  2 steps + save + restore into a fresh net + 2 steps == 4 uninterrupted steps, bit for bit."""
  rng = np.random.default_rng(3)
  batches = [[rng.uniform(size=(1, 256, 256, c)).astype(np.float32) for c in (6, 6, 3, 3)] for _ in range(4)]
  net_a, sess_a, nodes_a, ph_a = _train_net()
  init = {}
  for w in (0, 1, 2):
    init.update(net_a.engine.get_params(w))
  for b in batches:
    sess_a.run(nodes_a["Train_op"], feed_dict=dict(zip(ph_a, b)))
  net_b, sess_b, nodes_b, ph_b = _train_net()
  net_b.engine.load_params(init)
  for b in batches[:2]:
    sess_b.run(nodes_b["Train_op"], feed_dict=dict(zip(ph_b, b)))
  path = net_b.save(str(tmp_path / ("ck.npz" if fmt == "npz" else "ckpt_pixrefer/pixrefernet-4")))
  if fmt == "tf":   # a TensorFlow V2 checkpoint under the reference's names: Saver slots, beta powers, moving statistics
    from voicepuppet_amd.utils import tf_checkpoint
    names = tf_checkpoint.CheckpointReader(path).get_variable_to_shape_map()
    for k in ("generator/encoder_1/conv2d/kernel", "generator/encoder_1/conv2d/kernel/Adam_1", "generator_train/beta2_power",
              "discriminator/layer_2/batch_normalization/moving_variance", "vgg_16/conv3/conv3_3/weights", "global_step"):
      assert k in names, k
    path = str(tmp_path / "ckpt_pixrefer")      # restore through the directory's `checkpoint` state file
  net_c, sess_c, nodes_c, ph_c = _train_net()
  net_c.engine.load_params({k: v for k, v in init.items() if k.startswith("vgg_16")})
  net_c.restore(path)
  assert net_c.global_step == 4 and net_c.engine.t_g == 2 and net_c.engine.t_d == 2
  for b in batches[2:]:
    sess_c.run(nodes_c["Train_op"], feed_dict=dict(zip(ph_c, b)))
  assert net_c.global_step == net_a.global_step == 8
  for w in (0, 1):
    assert torch.equal(net_a.engine.arena(w), net_c.engine.arena(w))
    for k in (0, 1):
      key = "g" if w == 0 else "d"
      assert torch.equal(net_a.engine.adam[key][k], net_c.engine.adam[key][k])


def test_vgg_and_bfmnet_weights_come_from_tensorflow_checkpoints(tmp_path):
  """init_variables() reads allmodels/vgg_16.ckpt itself (pixrefer.py:325-327); BFMNet.restore reads 'ckpt_bfmnet/bfmnet-65000'
  (infer_bfmvid.py:217): TensorFlow bundles written here with known values, read back into the device arenas."""
  from voicepuppet_amd.bfmnet.bfmnet import BFMNet
  from voicepuppet_amd.pixrefer.pixrefer import PixReferNet
  from voicepuppet_amd.runtime import Placeholder, Session
  from voicepuppet_amd.utils import tf_checkpoint
  from oracle import pixrefer_ref as ref
  rng = np.random.default_rng(11)
  vgg = {n: rng.normal(0, 0.05, s).astype(np.float32) for n, s in ref.vgg_manifest()}
  vgg["vgg_16/fc8/biases"] = np.zeros(1000, np.float32)           # present in the real file, not used by the trunk
  vpath = tf_checkpoint.write_checkpoint(str(tmp_path / "allmodels" / "vgg_16.ckpt"), vgg)
  net = PixReferNet(CFG)
  p = net.params
  p.batch_size = 1
  p.ngf = p.ndf = 8
  p.add_hparam("is_training", True)
  p.sess = Session()
  p.vgg_model_path = vpath
  p.amd = dict(p.amd, dtype="f32")
  net.set_params(p)
  net.build_train_op(*[Placeholder([1, 256, 256, c], "x") for c in (6, 6, 3, 3)])
  got = net.engine.get_params(2)
  for n, _ in ref.vgg_manifest():
    np.testing.assert_array_equal(got[n], vgg[n])

  bfm = BFMNet(CFG)
  bp = bfm.params
  bp.batch_size = 1
  bfm.set_params(bp)
  bfm.build_inference_op(Placeholder([1, 5, 1], "ears"), Placeholder([1, 25, 80], "mfccs"), Placeholder([1], "seq"))
  w = {n: rng.normal(0, 0.05, s).astype(np.float32) for n, _, s in bfm.engine.manifest}
  w["global_step"] = np.int64(65000)
  bpath = tf_checkpoint.write_checkpoint(str(tmp_path / "ckpt_bfmnet" / "bfmnet-65000"), w)
  bfm.restore(bpath)
  back = bfm.engine.get_params()
  for n, _, _ in bfm.engine.manifest:
    np.testing.assert_array_equal(back[n], w[n])
  with pytest.raises(KeyError):
    bfm.restore(vpath)


def test_bfmnet_train_cli_checkpoints_and_restores(tmp_path, monkeypatch, capsys):
  """train_bfmnet.py end to end on synthetic clips: train / eval graphs, a TensorFlow-format checkpoint with the Adam slots, and a second
  run that resumes from it (train_bfmnet.py:93-98,141-145)."""
  from voicepuppet_amd.bfmnet import train_bfmnet
  from voicepuppet_amd.utils import tf_checkpoint
  monkeypatch.chdir(tmp_path)
  train_bfmnet.main(["--config_path", CFG, "--steps", "4", "--batch_size", "2", "--eval_step", "2", "--save_step", "2"])
  out = capsys.readouterr().out
  assert out.count("Step ") == 4 and out.count("Evaluation >>> Loss=") == 2
  assert os.path.exists("ckpt_bfmnet/bfmnet-4.index") and os.path.exists("ckpt_bfmnet/checkpoint")
  d = tf_checkpoint.read_checkpoint("ckpt_bfmnet")
  k = "bfm_coeff_decoder/dense_2/kernel"
  assert int(d["global_step"]) == 4 and k + "/Adam" in d and k + "/Adam_1" in d
  assert abs(float(d["beta1_power"]) - 0.9 ** 5) < 1e-6 and np.abs(d[k + "/Adam_1"]).max() > 0
  assert not any(n.endswith("moving_mean/Adam") for n in d)
  train_bfmnet.main(["--config_path", CFG, "--steps", "1", "--batch_size", "2"])
  out = capsys.readouterr().out
  assert "Restore from ckpt_bfmnet" in out and "Step 5:" in out


def test_bfmnet_build_train_op_learns_a_fixed_batch():
  """Feeding one fixed batch through Train_op lowers its own loss; Global_step / Lr / Grads / Tvars answer as the reference's nodes; the
  eval graph sees the trained variables; save -> restore reproduces the next step exactly (Adam state included)."""
  from voicepuppet_amd.bfmnet.bfmnet import BFMNet
  from voicepuppet_amd.runtime import Session
  rng = np.random.default_rng(0)
  B, T = 2, 24
  coeff = rng.normal(0, 0.5, (B, T, 257)).astype(np.float32)
  ears = rng.uniform(0.6, 0.9, (B, T, 1)).astype(np.float32)
  mfccs = rng.normal(0, 1, (B, 5 * T, 80)).astype(np.float32)
  seq = np.array([24, 17], np.int32)

  def make():
    net = BFMNet(CFG)
    p = net.params
    p.batch_size = B
    p.training = dict(p.training, drop_rate=0.0, learning_rate=1e-3)
    net.set_params(p)
    tr = net.build_train_op(coeff, ears, mfccs, seq)
    ev = net.build_eval_op(coeff, ears, mfccs, seq)
    return net, tr, ev
  net, tr, ev = make()
  net.train_engine.draw_masks = lambda *a, **k: None          # no dropout at all: the run is deterministic
  sess = Session()
  losses = []
  for i in range(12):
    _, loss, lr, gs = sess.run([tr["Train_op"], tr["Loss"], tr["Lr"], tr["Global_step"]])
    losses.append(float(loss))
    assert gs == i + 1 and abs(lr - 1e-3) < 1e-9
  assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0]
  grads, tvars = sess.run([tr["Grads"], tr["Tvars"]])
  assert len(grads) == len(tvars) == len(net.train_engine.trainables()) and all(g.shape == w.shape for g, w in zip(grads, tvars))
  assert sess.run(tr["Global_step"]) == 12                    # fetching Grads without Train_op applies nothing
  e1, pred = sess.run([ev["Loss"], ev["BFMCoeffDecoder"]])
  assert np.isfinite(e1) and pred.shape == (B, T, 64)
  import tempfile
  with tempfile.TemporaryDirectory() as d:
    path = net.save(os.path.join(d, "bfmnet-12"))
    net2, tr2, ev2 = make()
    net2.train_engine.draw_masks = lambda *a, **k: None
    net2.restore(path)
    assert net2.global_step == 12 and net2.train_engine.step_t == 12
    a = sess.run([tr["Train_op"], tr["Loss"]])[1]
    b = sess.run([tr2["Train_op"], tr2["Loss"]])[1]
    assert a == b
    wa, wb = net.train_engine.get_params(), net2.train_engine.get_params()
    assert all(np.array_equal(wa[k], wb[k]) for k in wa)


def test_bfmnet_restore_after_5000_steps_continues_bit_for_bit(tmp_path):
  """float32 0.9 ** (t + 1) is exactly 0 from t ~ 1000 on, the first checkpoint is written at step 5000
  (train_bfmnet.py:141) - restore must recover the Adam step count from beta2_power, in TF format, and the next step must be
  the one the uninterrupted run takes."""
  from voicepuppet_amd.bfmnet.bfmnet import BFMNet
  from voicepuppet_amd.runtime import Session
  from voicepuppet_amd.utils import tf_checkpoint
  rng = np.random.default_rng(3)
  B, T = 2, 24
  coeff = rng.normal(0, 0.5, (B, T, 257)).astype(np.float32)
  ears = rng.uniform(0.6, 0.9, (B, T, 1)).astype(np.float32)
  mfccs = rng.normal(0, 1, (B, 5 * T, 80)).astype(np.float32)
  seq = np.array([24, 20], np.int32)

  def make():
    net = BFMNet(CFG)
    p = net.params
    p.batch_size = B
    p.training = dict(p.training, drop_rate=0.0)
    net.set_params(p)
    tr = net.build_train_op(coeff, ears, mfccs, seq)
    net.train_engine.draw_masks = lambda *a, **k: None
    return net, tr
  sess = Session()
  net, tr = make()
  for _ in range(2):
    sess.run([tr["Train_op"]])
  net.train_engine.step_t = 5000          # as if 5000 updates had been applied (only the bias correction depends on it)
  net.global_step = 5000
  path = net.save(str(tmp_path / "bfmnet-5000"))
  d = tf_checkpoint.read_checkpoint(path)
  assert float(d["beta1_power"]) == 0.0 and 0.0 < float(d["beta2_power"]) < 1.0      # the situation the bug was about
  net2, tr2 = make()
  net2.restore(path)
  assert net2.train_engine.step_t == 5000 and net2.global_step == 5000
  a = sess.run([tr["Train_op"], tr["Loss"]])[1]
  b = sess.run([tr2["Train_op"], tr2["Loss"]])[1]
  assert a == b
  wa, wb = net.train_engine.get_params(), net2.train_engine.get_params()
  assert all(np.array_equal(wa[k], wb[k]) for k in wa)


def test_bfmnet_train_cli_on_clip_folders(tmp_path, monkeypatch, capsys):
  """train_bfmnet.py over real files in the reference's formats (folder list, audio.wav, landmark.txt, bfmcoeff.txt): silence trim,
  24-frame slices, log-mel on the device, two training steps and an evaluation."""
  import yaml
  from scipy.io import wavfile
  from voicepuppet_amd.bfmnet import train_bfmnet
  rng = np.random.default_rng(0)
  lines = []
  for k, frames in enumerate((60, 75)):
    folder = tmp_path / ("clip%d" % k)
    folder.mkdir()
    n = frames * 640
    t = np.arange(n) / 16000.0
    y = 0.4 * np.sin(2 * np.pi * (180 + 40 * k) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t))
    y[:3000] = 0
    wavfile.write(str(folder / "audio.wav"), 16000, (y * 32767).astype(np.int16))
    np.savetxt(str(folder / "bfmcoeff.txt"), rng.normal(0, 0.5, (frames, 257)), delimiter=",", fmt="%.6f")
    np.savetxt(str(folder / "landmark.txt"), rng.uniform(10, 200, (frames, 212)), delimiter=",", fmt="%.4f")
    lines.append("%s|%d" % (folder, frames))
  (tmp_path / "train.txt").write_text("\n".join(lines) + "\n")
  cfg = yaml.safe_load(open(CFG))
  cfg["default"]["train_dataset_path"] = str(tmp_path / "train.txt")
  cfg["default"]["eval_dataset_path"] = str(tmp_path / "train.txt")
  cfg["default"]["amd"]["synthetic_data"] = False          # the dataset must be read; the face model still falls back to the stand-in
  ypath = tmp_path / "params.yml"
  ypath.write_text(yaml.safe_dump(cfg))
  monkeypatch.chdir(tmp_path)
  with pytest.raises(IOError):                              # no stand-in face model allowed either -> says so
    train_bfmnet.main(["--config_path", str(ypath), "--steps", "1", "--batch_size", "2"])
  cfg["default"]["amd"]["synthetic_data"] = "auto"
  ypath.write_text(yaml.safe_dump(cfg))
  train_bfmnet.main(["--config_path", str(ypath), "--steps", "2", "--batch_size", "2", "--eval_step", "2", "--save_step", "100"])
  out = capsys.readouterr().out
  assert out.count("Step ") == 2 and out.count("Evaluation >>> Loss=") == 1
  losses = [float(l.split("Loss=")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Step ")]
  assert all(np.isfinite(losses))


def test_node_fetches_are_formed_by_the_library_kernel():
  """PixReferNet.execute's Outputs / Outputs_u8 / Alphas / Outputs_FG values come from vp_pixrefer_fetch (one kernel of the library per
  fetch): bit-equal to the reference's float32 expressions (pixrefer.py:284, 424, 436) evaluated with torch on the
  same device buffers, on a training and on an inference plan."""
  import torch
  from voicepuppet_amd.engine import PixReferEngine
  g = torch.Generator(device="cpu").manual_seed(1)
  for training in (True, False):
    eng = PixReferEngine(2, 256, 8, 8, dtype="bf16", training=training)
    eng.load_params(eng.random_params(3))
    batch = [torch.rand(2, 256, 256, c, generator=g).cuda() for c in ((6, 6, 3, 3) if training else (6, 3, 3))]
    eng.forward(*batch)
    torch.cuda.synchronize()
    raw, fg, o4 = eng.tensor("Outputs_raw"), eng.tensor("Outputs_FG"), eng.tensor("gen_out4")
    alpha = (o4[..., 3:] + 1) / 2
    want = {"Outputs": (raw + 1) / 2, "Outputs_u8": ((raw + 1) / 2).clamp(0, 1).mul(255).to(torch.uint8), "Alphas": alpha.repeat(1, 1, 1, 3),
            "Outputs_FG": fg if training else ((fg + alpha - 1) + 1) / 2}
    for k, w in want.items():
      got = eng.fetch(k)
      assert got.dtype == w.dtype and got.shape == w.shape, k
      assert torch.equal(got, w), (k, training, float((got.float() - w.float()).abs().max()))


def test_host_fetch_path_holds_no_framework_arithmetic():
  """The per-frame fetch path of infer_bfmvid (PixReferNet.execute) and the scalar tail of the BFMNet training step hold no torch
  elementwise / reduction call any more: source-level guard next to test_training_step_holds_no_vendor_gemm."""
  import re
  root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "voicepuppet_amd")
  src = open(os.path.join(root, "pixrefer", "pixrefer.py")).read()
  body = src[src.index("  def execute(self, names, feed_dict):"):src.index("  # ---- checkpoints")]
  for pat in (r"\+ 1\) / 2", r"\.clamp", r"\.repeat\(", r"\.mul_\("):
    assert not re.search(pat, body), pat
  eng = open(os.path.join(root, "bfmnet", "train_engine.py")).read()
  for pat in (r"part\.sum\(\)", r"torch\.stack\(", r"torch\.sqrt\(ss\)", r"grads\.mul_\(", r"torch\.clamp\("):
    assert not re.search(pat, eng), pat


def test_infer_bfmvid_frame_batch_runs_no_framework_kernel_but_copies():
  """a frame batch of infer_bfmvid (build_inference_op fed a 3-channel FGInputs, infer_bfmvid.py:202-205, fetching the
  uint8 frames and Outputs_FG) traced op by op: every ATen call the host path makes is a copy / allocation / view - torch.cat and
  zeros_like on the foreground reference are gone (vp_pixrefer_forward_fg3 reads the [N,H,H,3] tensor as it is) - and the result equals
  the 6-channel feed of the same data bit for bit."""
  import torch
  from torch.utils._python_dispatch import TorchDispatchMode
  from voicepuppet_amd.pixrefer.pixrefer import PixReferNet
  from voicepuppet_amd.runtime import Placeholder

  net = PixReferNet(CFG)
  params = net.params
  params.batch_size = 2
  params.add_hparam('is_training', False)
  params.ngf = params.ndf = 8
  net.set_params(params)
  ih, fh, th = Placeholder((None, 256, 256, 6), 'inputs'), Placeholder((None, 256, 256, 3), 'fg'), Placeholder((None, 256, 256, 3), 'targets')
  nodes = net.build_inference_op(ih, fh, th)
  rng = np.random.default_rng(5)
  feed = {ih: rng.uniform(size=(2, 256, 256, 6)).astype(np.float32), fh: rng.uniform(size=(2, 256, 256, 3)).astype(np.float32),
          th: rng.uniform(size=(2, 256, 256, 3)).astype(np.float32)}
  net.execute(['Outputs_u8'], feed)       # (first call: packs the weights)

  seen = []

  class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
      seen.append(func.__name__ if hasattr(func, '__name__') else str(func))
      return func(*args, **(kwargs or {}))

  with Trace():
    got = net.execute(['Outputs_u8', 'Outputs_FG', 'Alphas'], feed)
  allowed = ('_to_copy', 'copy_', 'empty', 'empty_strided', 'lift_fresh', 'detach', 'alias', 'view', '_unsafe_view', 'as_strided', 'slice', 'select',
             'contiguous', 'clone', '_local_scalar_dense', 'unsqueeze', 'reshape', '_reshape_alias', 'to', 'cpu', 'is_pinned', '_pin_memory', 'record_stream')
  bad = sorted({n for n in seen if n.split('.')[0] not in allowed})
  assert not bad, (bad, seen)
  assert seen, "the trace saw nothing"
  # the same frames through the 6-channel entry point (what engine.forward did with torch.cat / zeros_like before)
  eng = net.engine
  fg6 = np.concatenate([feed[fh], np.zeros_like(feed[fh])], axis=-1)
  eng.forward(torch.tensor(feed[ih]).cuda(), torch.tensor(fg6).cuda(), torch.tensor(feed[th]).cuda())
  want = eng.fetch('Outputs_u8').cpu().numpy()
  assert np.array_equal(got['Outputs_u8'], want)
