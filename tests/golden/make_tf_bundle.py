"""Writes the TensorFlow-checkpoint fixtures under tests/golden/tf_bundle/ byte by byte, WITHOUT using the product's reader or
writer (voicepuppet_amd/utils/tf_checkpoint.py) - an independent statement of the on-disk format, so the reader is not only
tested against its own writer.  TensorFlow cannot run in the build container; the layout below follows the published formats:

  V2 bundle  (tensorflow/core/util/tensor_bundle/tensor_bundle.cc, tensorflow/core/protobuf/tensor_bundle.proto)
  SSTable    (tensorflow/core/lib/io/format.cc, block_builder.cc, table_builder.cc == the LevelDB table format)
  V1 file    (tensorflow/core/util/tensor_slice_writer.cc, saved_tensor_slice.proto)

Deliberately awkward choices a real file may contain: two data shards, several data blocks in the index, restart interval 2
(so most keys are prefix-compressed), a key that is a prefix of the next one, int32 / int64 / bfloat16 / scalar tensors,
non-zero offsets, a snappy-compressed block, and (V1) both packed float_val and tensor_content payloads.

    python tests/golden/make_tf_bundle.py      # rewrites tests/golden/tf_bundle/*
"""
import os
import struct

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_bundle")
MAGIC = 0xdb4775248b80fb57


def crc32c(data):
  c = 0xFFFFFFFF
  for b in data:
    c ^= b
    for _ in range(8):
      c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
  return c ^ 0xFFFFFFFF


def masked(c):
  return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def vi(v):
  v &= (1 << 64) - 1
  o = bytearray()
  while v >= 0x80:
    o.append((v & 0x7F) | 0x80)
    v >>= 7
  o.append(v)
  return bytes(o)


def ld(field, payload):      # length-delimited protobuf field
  return vi((field << 3) | 2) + vi(len(payload)) + payload


def shape_proto(shape):
  return b"".join(ld(2, vi(8) + vi(d)) for d in shape)          # repeated Dim dim = 2 { int64 size = 1 }


def block(records, restart_interval):
  """records: [(key, value)] sorted.  LevelDB block: shared | non_shared | value_len | key delta | value ... restarts, count."""
  buf, restarts, last = bytearray(), [], b""
  for i, (k, v) in enumerate(records):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(buf))
    else:
      while shared < min(len(k), len(last)) and k[shared] == last[shared]:
        shared += 1
    buf += vi(shared) + vi(len(k) - shared) + vi(len(v)) + k[shared:] + v
    last = k
  if not restarts:
    restarts = [0]
  return bytes(buf) + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))


def snappy_literal_only(raw):
  """A valid snappy stream made of literal elements only (no back references): what a compressor may emit for incompressible data."""
  out = bytearray(vi(len(raw)))
  for i in range(0, len(raw), 60):
    piece = raw[i:i + 60]
    out.append((len(piece) - 1) << 2)
    out += piece
  return bytes(out)


def table(blocks_of_records, restart_interval, compress_block=None):
  out, handles = bytearray(), []
  for bi, recs in enumerate(blocks_of_records):
    body = block(recs, restart_interval)
    ctype = 0
    if compress_block == bi:
      body, ctype = snappy_literal_only(body), 1
    off = len(out)
    out += body + bytes([ctype]) + struct.pack("<I", masked(crc32c(body + bytes([ctype]))))
    handles.append((recs[-1][0], vi(off) + vi(len(body))))

  def raw_block(body):
    off = len(out)
    out.extend(body + b"\x00" + struct.pack("<I", masked(crc32c(body + b"\x00"))))
    return vi(off) + vi(len(body))
  meta = raw_block(block([], 16))
  # index keys: any separator >= the block's last key and < the next block's first key; TF uses a shortened separator
  idx = raw_block(block([(k + (b"\x00" if i + 1 < len(handles) else b""), h) for i, (k, h) in enumerate(handles)], 1))
  footer = meta + idx
  out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
  return bytes(out)


def tensors():
  rng = np.random.default_rng(2024)
  t = {
      "generator/encoder_1/conv2d/kernel": rng.normal(0, 0.02, (4, 4, 6, 8)).astype(np.float32),
      "generator/encoder_1/conv2d/kernel/Adam": rng.normal(0, 1e-3, (4, 4, 6, 8)).astype(np.float32),
      "generator/encoder_1/conv2d/kernel/Adam_1": rng.uniform(0, 1e-5, (4, 4, 6, 8)).astype(np.float32),
      "generator/encoder_1/conv2d/bias": np.zeros(8, np.float32),
      "generator/encoder_2/batch_normalization/gamma": rng.normal(1, 0.02, 16).astype(np.float32),
      "generator/encoder_2/batch_normalization/moving_variance": np.ones(16, np.float32),
      "discriminator_train/beta1_power": np.float32(0.5 ** 8),
      "discriminator_train/beta2_power": np.float32(0.999 ** 8),
      "generator_train/beta2_power": np.float32(0.999 ** 8),
      "global_step": np.int32(14),
      "step64": np.int64(-3),
      "vgg_16/conv1/conv1_1/weights": rng.normal(0, 0.1, (3, 3, 3, 64)).astype(np.float32),
      "vgg_16/conv1/conv1_1/biases": rng.normal(0, 0.1, 64).astype(np.float32),
      "vgg_16/mean_rgb": np.asarray([123.68, 116.78, 103.94], np.float32),
      "half/bf16_vector": (rng.normal(0, 1, 10).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16),   # stored as DT_BFLOAT16
  }
  return t


DT = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def write_v2():
  t = tensors()
  names = sorted(t, key=lambda s: s.encode())
  shards = [bytearray(b"\xAA" * 24), bytearray()]          # shard 0 starts with padding: offsets are not zero
  recs = [(b"", vi(8) + vi(2) + ld(3, vi(8) + vi(1)))]      # BundleHeaderProto: num_shards = 2, version { producer = 1 }
  expect = {}
  for i, n in enumerate(names):
    a = t[n]
    sid = i % 2
    raw = np.ascontiguousarray(a).tobytes()
    dtype = 14 if n == "half/bf16_vector" else DT[a.dtype]
    off = len(shards[sid])
    shards[sid] += raw + b"\x55" * (i % 3)                  # gaps between tensors
    e = vi(8) + vi(dtype) + ld(2, shape_proto(a.shape))
    if sid:
      e += vi(0x18) + vi(sid)
    e += vi(0x20) + vi(off) + vi(0x28) + vi(len(raw)) + vi(0x35) + struct.pack("<I", masked(crc32c(raw)))
    recs.append((n.encode(), e))
    expect[n] = ((a.astype(np.uint32) << 16).view(np.float32) if dtype == 14 else a)
  blocks = [recs[0:5], recs[5:6], recs[6:12], recs[12:]]
  os.makedirs(OUT, exist_ok=True)
  with open(os.path.join(OUT, "model.ckpt-14.index"), "wb") as f:
    f.write(table(blocks, restart_interval=2, compress_block=2))
  for sid in range(2):
    with open(os.path.join(OUT, "model.ckpt-14.data-%05d-of-00002" % sid), "wb") as f:
      f.write(bytes(shards[sid]))
  with open(os.path.join(OUT, "checkpoint"), "w") as f:
    f.write('model_checkpoint_path: "model.ckpt-14"\nall_model_checkpoint_paths: "model.ckpt-7"\nall_model_checkpoint_paths: "model.ckpt-14"\n')
  np.savez(os.path.join(OUT, "expected.npz"), **{k.replace("/", "|"): v for k, v in expect.items()})


def write_v1():
  """One-file checkpoint: key "" -> SavedTensorSlices{meta}, other keys -> SavedTensorSlices{data{name, slice, data: TensorProto}}."""
  rng = np.random.default_rng(7)
  w = rng.normal(0, 0.1, (3, 3, 3, 4)).astype(np.float32)
  b = rng.normal(0, 0.1, 4).astype(np.float32)
  gs = np.asarray(123, np.int64)

  def tensor_proto(a, as_content):
    # TensorSliceWriter::SaveData -> Fill<T> sets ONLY the typed *_val field: no dtype, no tensor_shape in a slice's TensorProto
    # (they live in the meta record under the empty key).  The tensor_content variant (another writer) carries its own header.
    if as_content:
      return vi(8) + vi(DT[a.dtype]) + ld(2, shape_proto(a.shape)) + ld(4, a.tobytes())
    if a.dtype == np.float32:
      return ld(5, a.tobytes())                              # packed repeated float float_val = 5
    return ld(10, b"".join(vi(int(x)) for x in a.reshape(-1)))       # packed repeated int64 int64_val = 10

  def rec(name, a, as_content):
    full_slice = b"".join(ld(1, b"") for _ in a.shape)       # TensorSliceProto: one Extent per dim, empty = full
    saved = ld(1, name.encode()) + ld(2, full_slice) + ld(3, tensor_proto(a, as_content))
    return ld(2, saved)
  items = {"vgg_16/conv1/conv1_1/weights": (w, False), "vgg_16/conv1/conv1_1/biases": (b, True), "global_step": (gs, False)}
  meta = ld(1, b"".join(ld(1, ld(1, n.encode()) + ld(2, shape_proto(a.shape)) + vi(0x18) + vi(DT[a.dtype])) for n, (a, _) in items.items()))
  # the real keys are an ordered-code encoding of (name, slice); the reader only needs them sorted and unique
  recs = [(b"", meta)] + [(b"\x00" + n.encode() + b"\x00\x01", rec(n, a, c)) for n, (a, c) in sorted(items.items())]
  with open(os.path.join(OUT, "v1_model.ckpt"), "wb") as f:
    f.write(table([recs[:2], recs[2:]], restart_interval=16))
  np.savez(os.path.join(OUT, "expected_v1.npz"), **{k.replace("/", "|"): a for k, (a, _) in items.items()})


if __name__ == "__main__":
  write_v2()
  write_v1()
  print("wrote", sorted(os.listdir(OUT)))
