"""TensorFlow checkpoint importer / exporter (voicepuppet_amd/utils/tf_checkpoint.py) against fixtures assembled byte by byte by
tests/golden/make_tf_bundle.py (an independent statement of the V2 bundle / SSTable / V1 formats: two shards, several index
blocks, prefix-compressed keys, a snappy block, bf16 / int32 / int64 / scalar entries), plus write -> read round trips."""
import os

import numpy as np
import pytest

from voicepuppet_amd.utils import tf_checkpoint as tc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_bundle")


def _expected(name):
  z = np.load(os.path.join(HERE, name))
  return {k.replace("|", "/"): z[k] for k in z.files}


def test_crc32c_known_answers_and_lane_parallel_form():
  assert tc.crc32c(b"123456789") == 0xE3069283                 # the CRC-32C check value (RFC 3720 B.4)
  assert tc.crc32c(b"\x00" * 32) == 0x8A9136AA and tc.crc32c(b"\xff" * 32) == 0x62A8AB43
  raw = np.random.default_rng(0).integers(0, 256, 200_003, dtype=np.uint8).tobytes()
  assert tc.crc32c_fast(raw) == tc.crc32c(raw)
  assert tc.crc32c(raw[100:], tc.crc32c(raw[:100])) == tc.crc32c(raw)


def test_reads_the_hand_built_v2_bundle():
  want = _expected("expected.npz")
  r = tc.CheckpointReader(os.path.join(HERE, "model.ckpt-14"))
  assert r.num_shards == 2 and sorted(r.entries) == sorted(want)
  shapes = r.get_variable_to_shape_map()
  for k, v in want.items():
    got = r.get_tensor(k)
    assert got.dtype == v.dtype and got.shape == v.shape and list(shapes[k]) == list(v.shape), k
    np.testing.assert_array_equal(got, v)
  # directory form goes through the `checkpoint` state file; names= filters
  d = tc.read_checkpoint(HERE, names=["global_step", "vgg_16/conv1/conv1_1/weights", "not/there"])
  assert sorted(d) == ["global_step", "vgg_16/conv1/conv1_1/weights"] and int(d["global_step"]) == 14
  assert tc.latest_checkpoint(HERE).endswith("model.ckpt-14")
  assert tc.is_tf_checkpoint(os.path.join(HERE, "model.ckpt-14")) and not tc.is_tf_checkpoint(os.path.join(HERE, "expected.npz"))


def test_reads_the_hand_built_v1_file():
  want = _expected("expected_v1.npz")
  path = os.path.join(HERE, "v1_model.ckpt")
  assert tc.is_tf_checkpoint(path)
  got = tc.read_checkpoint(path)
  assert sorted(got) == sorted(want)
  for k, v in want.items():
    np.testing.assert_array_equal(got[k], v)
    assert got[k].shape == v.shape


def test_corruption_is_detected(tmp_path):
  import shutil
  for f in os.listdir(HERE):
    shutil.copy(os.path.join(HERE, f), tmp_path / f)
  p = tmp_path / "model.ckpt-14.data-00000-of-00002"
  b = bytearray(p.read_bytes())
  b[40] ^= 0x10
  p.write_bytes(bytes(b))
  r = tc.CheckpointReader(str(tmp_path / "model.ckpt-14"))
  bad = 0
  for k in r.entries:
    try:
      r.get_tensor(k)
    except ValueError:
      bad += 1
  assert bad == 1
  idx = tmp_path / "model.ckpt-14.index"
  b = bytearray(idx.read_bytes())
  b[10] ^= 0x01
  idx.write_bytes(bytes(b))
  with pytest.raises(ValueError):
    tc.CheckpointReader(str(tmp_path / "model.ckpt-14"))
  with pytest.raises(FileNotFoundError):
    tc.read_checkpoint(str(tmp_path / "nothing-here"))


def test_write_then_read_round_trip_many_blocks(tmp_path):
  rng = np.random.default_rng(1)
  t = {"scope_%03d/layer/kernel" % i: rng.normal(size=(3, i % 5 + 1, 2)).astype(np.float32) for i in range(300)}   # > one 4 KB index block
  t["scope_000/layer/kernel/Adam"] = rng.normal(size=(3, 1, 2)).astype(np.float32)
  t["global_step"] = np.int32(20000)
  t["big"] = rng.normal(size=(70_000,)).astype(np.float32)      # > 64 KB: the lane-parallel CRC path on both sides
  prefix = tc.write_checkpoint(str(tmp_path / "ckpt" / "net-20000"), t)
  assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
  assert tc.latest_checkpoint(str(tmp_path / "ckpt")) == prefix
  got = tc.read_checkpoint(str(tmp_path / "ckpt"))
  assert sorted(got) == sorted(t)
  for k in t:
    np.testing.assert_array_equal(got[k], t[k])
    assert got[k].dtype == np.asarray(t[k]).dtype
  # and the independent fixture's table layer parses what the product writer produced (keys sorted, handles consistent)
  keys = [k for k, _ in tc.read_table(prefix + ".index")]
  assert keys == sorted(keys) and keys[0] == b""


def test_adam_step_count_from_saved_beta_powers():
  for t in (0, 1, 7, 149, 20000):
    b1p, b2p = np.float32(0.5 ** (t + 1)), np.float32(0.999 ** (t + 1))
    assert tc.adam_steps_from_beta_powers(b1p, b2p, 0.5, 0.999) == t
  assert tc.adam_steps_from_beta_powers(0.0, 0.0, 0.5, 0.999) >= 1000000     # both underflowed: bias correction is 1
