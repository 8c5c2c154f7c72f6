"""CPU checks of BFMNetDataGenerator's host half (generator/generator.py:377-481 of the reference): the silence trim, the 24-frame slicing
with its PCM window arithmetic, the list/file formats.  The log-mel step (process_data) runs on the device and is covered by -m gpu."""
import os

import numpy as np

from oracle import audio_ref as ar
from voicepuppet_amd.generator.generator import BFMNetDataGenerator, first_nonsilent_sample

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config", "params.yml")


def _frame_db(y, frame_length=2048, hop=512):
  """Literal per-frame loop of the published librosa.effects.split pre-processing (reflect-padded centred frames)."""
  yp = np.pad(y, frame_length // 2, mode="reflect")
  n = 1 + (len(yp) - frame_length) // hop
  mse = np.array([np.mean(yp[i * hop:i * hop + frame_length].astype(np.float64) ** 2) for i in range(n)])
  return 10 * np.log10(np.maximum(1e-10, mse)) - 10 * np.log10(np.maximum(1e-10, mse.max()))


def test_first_nonsilent_sample_matches_frame_loop():
  rng = np.random.default_rng(0)
  sr = 16000
  for lead in (0, 3000, 9000, 20000):
    t = np.arange(sr) / sr
    y = np.concatenate([1e-4 * rng.normal(size=lead), 0.4 * np.sin(2 * np.pi * 220 * t) * np.minimum(1, t * 8)]).astype(np.float32)
    db = _frame_db(y)
    want = min(int(np.flatnonzero(db > -20)[0]) * 512, len(y))
    got = first_nonsilent_sample(y, top_db=20)
    assert got == want
    assert got <= lead + 4000 and (lead == 0 or got > lead - 2048)
  assert first_nonsilent_sample(np.zeros(100, np.float32)) == 0


def _write_clip(folder, frames, lead_silence, rng):
  from scipy.io import wavfile
  os.makedirs(folder)
  sr = 16000
  n = int(frames * sr / 25)
  t = np.arange(n) / sr
  y = 0.4 * np.sin(2 * np.pi * 200 * t)
  y[:lead_silence] = 0
  wavfile.write(os.path.join(folder, "audio.wav"), sr, (y * 32767).astype(np.int16))
  coeff = rng.normal(0, 1, (frames, 257)).astype(np.float32)
  np.savetxt(os.path.join(folder, "bfmcoeff.txt"), coeff, delimiter=",", fmt="%.6f")
  lm = rng.uniform(10, 200, (frames, 212)).astype(np.float32)
  np.savetxt(os.path.join(folder, "landmark.txt"), lm, delimiter=",", fmt="%.4f")
  return coeff, lm


def test_slices_of_a_clip_folder(tmp_path):
  rng = np.random.default_rng(1)
  frames, lead = 80, 16000 * 12 // 25 + 100           # a little over 12 video frames of leading silence
  coeff, lm = _write_clip(str(tmp_path / "clipA"), frames, lead, rng)
  _write_clip(str(tmp_path / "short"), 10, 0, rng)      # fewer than 24 frames: yields nothing
  lst = tmp_path / "train.txt"
  lst.write_text("%s|%d\n%s|%d\n%s|5\n" % (tmp_path / "clipA", frames, tmp_path / "short", 10, tmp_path / "missing"))
  gen = BFMNetDataGenerator(CFG)
  p = gen.params
  p.dataset_path = str(lst)
  p.amd = dict(p.amd, synthetic_data=False)
  gen.set_params(p)
  got = list(gen.iterator())
  from voicepuppet_amd.generator.loader import WavLoader
  pcm = WavLoader(sr=16000).get_data(str(tmp_path / "clipA" / "audio.wav"))
  start = first_nonsilent_sample(pcm)
  skip = int(start // 640)
  assert 11 <= skip <= 13
  assert len(got) == (frames - skip) // 24
  ear_all = 1 - gen.ear_compute(lm)
  for i, (c, e, w, n) in enumerate(got):
    assert c.shape == (24, 257) and e.shape == (24, 1) and n == 24
    # identity coefficients: the mean over the trimmed clip, the same on every frame; the rest untouched
    assert np.allclose(c[:, :80], coeff[skip:, :80].mean(0, keepdims=True), atol=1e-5)
    assert np.allclose(c[:, 80:], coeff[skip + 24 * i: skip + 24 * (i + 1), 80:], atol=1e-5)
    # reference quirk (generator.py:471-472): `ear` is sliced from the un-trimmed start
    assert np.allclose(e, ear_all[24 * i:24 * (i + 1)], atol=1e-5)
    # the PCM window gives exactly 24 * 5 log-mel rows and starts 24 * i video frames after the trim point
    assert w.shape == (128 * (24 * 5 - 1) + 512,)
    assert ar.extract_mfcc(w[None].astype(np.float64)).shape[1] == 120
    seg = pcm[start + i * 24 * 640: start + i * 24 * 640 + w.shape[0]]
    assert np.array_equal(w[:seg.shape[0]], seg) and not w[seg.shape[0]:].any()


def test_synthetic_fallback_shapes():
  gen = BFMNetDataGenerator(CFG)
  p = gen.params
  p.dataset_path = "/nonexistent/train.txt"
  gen.set_params(p)
  c, e, w, n = next(gen.iterator())
  assert c.shape == (24, 257) and e.shape == (24, 1) and w.shape == (gen.pcm_length(24),) and n == 24
  assert np.array_equal(c[:, :80], np.repeat(c[:1, :80], 24, 0))


def test_background_batch_thread_is_reproducible_and_stops(monkeypatch):
  """ADVICE r3 on _MfccIterator: the batch thread draws from a private random.Random seeded from the module-level state on the
  caller's thread (so the sample order is a function of random.seed() even while the main thread draws), and it stops when the
  iterator is closed or dropped.  process_data (log-mel, device) is replaced by the identity: this is the host half only."""
  import gc
  import random
  import threading
  import time
  monkeypatch.setattr(BFMNetDataGenerator, "process_data", lambda self, c, e, p, n: (c, e, p, n))

  def run(noise):
    random.seed(123)
    g = BFMNetDataGenerator(CFG)
    prm = g.params
    prm.dataset_path = "/nonexistent/train.txt"          # -> synthetic clips
    prm.batch_size = 3
    prm.shuffle_bufsize = 1                              # (the shuffle buffer itself draws from an unseeded numpy generator, as tf.data does)
    g.set_params(prm)
    it = g.get_dataset().make_one_shot_iterator()
    out = []
    for k in range(4):
      if noise:
        random.random()                                   # the main thread draws between batches: must not change the order
      out.append(it.next_batch()[0].copy())
    return it, out
  it1, a = run(False)
  it2, b = run(True)
  for x, y in zip(a, b):
    np.testing.assert_array_equal(x, y)
  names = lambda: [t for t in threading.enumerate() if t.name == "bfmnet-batches" and t.is_alive()]
  assert len(names()) == 2
  it1.close()
  assert len(names()) == 1
  del it2, a, b
  gc.collect()
  t0 = time.time()
  while names() and time.time() - t0 < 5:
    time.sleep(0.1)
  assert not names()


def test_two_iterators_over_one_generator_do_not_share_draws(monkeypatch):
  """the private random.Random lives on the ITERATOR (not on the shared DataGenerator): a second iterator over the same
  owner neither inherits nor overwrites the first one's generator, and making an iterator leaves the module-level stream of a caller
  that seeded it untouched."""
  import random
  monkeypatch.setattr(BFMNetDataGenerator, "process_data", lambda self, c, e, p, n: (c, e, p, n))
  random.seed(7)
  g = BFMNetDataGenerator(CFG)
  prm = g.params
  prm.dataset_path = "/nonexistent/train.txt"
  prm.batch_size = 2
  prm.shuffle_bufsize = 1
  g.set_params(prm)
  ds = g.get_dataset()
  state = random.getstate()
  it1 = ds.make_one_shot_iterator()
  assert random.getstate() == state                      # no draw from the caller's stream
  first = it1.next_batch()[0].copy()
  it2 = ds.make_one_shot_iterator()                      # same module state -> same seed -> the same clips, from its OWN generator
  second = it2.next_batch()[0].copy()
  np.testing.assert_array_equal(first, second)
  assert it1._rand is not it2._rand and not hasattr(g, "_private_random")
  nxt1, nxt2 = it1.next_batch()[0], it2.next_batch()[0]  # both continue their own sequence
  np.testing.assert_array_equal(nxt1, nxt2)
  it1.close(); it2.close()


def test_sample_order_is_a_function_of_the_seed_across_processes():
  """two PROCESSES that call random.seed(5) see the same first batch (the iterator's private generator was seeded with
  hash(random.getstate()), which contains hash(None) - an address before CPython 3.12 - so every process drew a different order)."""
  import subprocess
  import sys
  code = r"""
import random, sys, zlib
import numpy as np
sys.path.insert(0, %r)
from voicepuppet_amd.generator.generator import BFMNetDataGenerator
BFMNetDataGenerator.process_data = lambda self, c, e, p, n: (c, e, p, n)
random.seed(5)
g = BFMNetDataGenerator(%r)
prm = g.params
prm.dataset_path = "/nonexistent/train.txt"
prm.batch_size = 3
prm.shuffle_bufsize = 1
g.set_params(prm)
it = g.get_dataset().make_one_shot_iterator()
b = it.next_batch()
print("CRC", zlib.crc32(np.ascontiguousarray(b[0]).tobytes()), zlib.crc32(np.ascontiguousarray(b[2]).tobytes()), it._rand.getrandbits(32))
it.close()
""" % (ROOT, CFG)
  outs = []
  for hashseed in ("1", "2"):                 # (different string-hash seeds too: nothing of the order may hang on hash())
    env = dict(os.environ, PYTHONHASHSEED=hashseed)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append([l for l in r.stdout.splitlines() if l.startswith("CRC")][0])
  assert outs[0] == outs[1], outs
