"""TensorFlow checkpoint (tf.train.Saver V2 "tensor bundle") reader and writer in plain Python + numpy - no TensorFlow.

The reference restores three external checkpoints through tf.train.Saver / slim.assign_from_checkpoint_fn:
  ckpt_bfmnet/bfmnet-65000, ckpt_pixrefer/pixrefernet-20000  (voicepuppet/pixrefer/infer_bfmvid.py:207-218)
  allmodels/vgg_16.ckpt                                       (voicepuppet/pixrefer/pixrefer.py:325-327; a V1 or V2 file)
and saves 'ckpt_pixrefer/pixrefernet-<global_step>' (voicepuppet/pixrefer/train_pixrefer.py:150).  The HIP executors keep their
parameters under the TF variable names, so {name: array} from here goes straight into PixReferEngine.load_params / load_adam
and BFMNetEngine.load_params.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*, restated from the published format):
  <prefix>.index                  an SSTable (LevelDB table format): sorted key -> value records in prefix-compressed blocks,
                                  each block followed by a 1-byte compression type and a masked CRC32C; a metaindex block, an index
                                  block (last key of each data block -> BlockHandle) and a 48-byte footer ending in the magic
                                  0xdb4775248b80fb57.  Key "" -> BundleHeaderProto {num_shards=1, endianness=2, version=3};
                                  key <tensor name> -> BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
  <prefix>.data-SSSSS-of-NNNNN    raw little-endian tensor bytes at (shard_id, offset, size).
A V1 checkpoint (one file, e.g. the slim vgg_16.ckpt download) is the same SSTable with SavedTensorSlices protos as values;
`read_checkpoint` reads that form too (full-tensor slices, float / int32 / int64 data).

UNPINNED BY TENSORFLOW: this module has never read a file TensorFlow wrote, nor has TensorFlow read one it wrote.  The three
checkpoints above are external downloads, TensorFlow is not installable in the build container and the reference ships no bundle.
What pins it: the published format restated above, CRC-32C known answers, and fixtures that tests/golden/make_tf_bundle.py encodes
byte by byte from that description with its own protobuf / varint / block writer (a SECOND encoding by the same author - exactly how
a misread of the V1 slice proto hid for a round).  Treat "reads V1 / V2" as "reads what the format description says V1 / V2 is" until
a real `vgg_16.ckpt` or `pixrefernet-20000` has been through it.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_BFLOAT16 = 14
_DT_OF = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- CRC32C (Castagnoli), masked the LevelDB way ---------------------------------------------------------------------------
def _crc_table():
  tab = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    tab.append(c)
  return tab


_CRC = _crc_table()


def crc32c(data, crc=0):
  c = crc ^ 0xFFFFFFFF
  for b in bytes(data):
    c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


def mask_crc(crc):
  return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- protobuf wire format (only what the bundle protos need) -----------------------------------------------------------------
def _varint(buf, pos):
  out = shift = 0
  while True:
    b = buf[pos]
    pos += 1
    out |= (b & 0x7F) << shift
    if not b & 0x80:
      return out, pos
    shift += 7


def _put_varint(v):
  out = bytearray()
  v &= (1 << 64) - 1
  while True:
    b = v & 0x7F
    v >>= 7
    if v:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _fields(buf):
  """[(field number, wire type, value)] of one message; length-delimited values stay bytes."""
  pos, out = 0, []
  buf = bytes(buf)
  while pos < len(buf):
    key, pos = _varint(buf, pos)
    fn, wt = key >> 3, key & 7
    if wt == 0:
      v, pos = _varint(buf, pos)
    elif wt == 1:
      v = buf[pos:pos + 8]
      pos += 8
    elif wt == 2:
      n, pos = _varint(buf, pos)
      v = buf[pos:pos + n]
      pos += n
    elif wt == 5:
      v = buf[pos:pos + 4]
      pos += 4
    else:
      raise ValueError("unsupported protobuf wire type %d" % wt)
    out.append((fn, wt, v))
  return out


def _signed(v):
  return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
  dims = []
  for fn, wt, v in _fields(buf):
    if fn == 2 and wt == 2:                       # TensorShapeProto.Dim
      size = 0
      for f2, w2, v2 in _fields(v):
        if f2 == 1 and w2 == 0:
          size = _signed(v2)
      dims.append(size)
    elif fn == 3 and wt == 0 and v:
      raise ValueError("tensor of unknown rank in a checkpoint")
  return tuple(dims)


def _shape_proto(shape):
  out = b""
  for d in shape:
    dim = b"\x08" + _put_varint(int(d))
    out += b"\x12" + _put_varint(len(dim)) + dim
  return out


def _parse_entry(buf):
  e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
  for fn, wt, v in _fields(buf):
    if fn == 1 and wt == 0:
      e["dtype"] = v
    elif fn == 2 and wt == 2:
      e["shape"] = _parse_shape(v)
    elif fn == 3 and wt == 0:
      e["shard_id"] = v
    elif fn == 4 and wt == 0:
      e["offset"] = v
    elif fn == 5 and wt == 0:
      e["size"] = v
    elif fn == 6 and wt == 5:
      e["crc32c"] = struct.unpack("<I", v)[0]
    elif fn == 7:
      e["sliced"] = True
  return e


def _entry_proto(dtype, shape, shard_id, offset, size, crc):
  out = b"\x08" + _put_varint(dtype)
  sp = _shape_proto(shape)
  out += b"\x12" + _put_varint(len(sp)) + sp
  if shard_id:
    out += b"\x18" + _put_varint(shard_id)
  if offset:
    out += b"\x20" + _put_varint(offset)
  out += b"\x28" + _put_varint(size)
  out += b"\x35" + struct.pack("<I", crc)
  return out


# ---- snappy (an index written with compression on; Saver writes it uncompressed) -------------------------------------------
def _snappy_decompress(buf):
  n, pos = _varint(buf, 0)
  out = bytearray()
  while pos < len(buf):
    tag = buf[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:
      ln = tag >> 2
      if ln >= 60:
        nb = ln - 59
        ln = int.from_bytes(buf[pos:pos + nb], "little")
        pos += nb
      ln += 1
      out += buf[pos:pos + ln]
      pos += ln
      continue
    if kind == 1:
      ln = ((tag >> 2) & 7) + 4
      off = ((tag >> 5) << 8) | buf[pos]
      pos += 1
    elif kind == 2:
      ln = (tag >> 2) + 1
      off = buf[pos] | (buf[pos + 1] << 8)
      pos += 2
    else:
      ln = (tag >> 2) + 1
      off = int.from_bytes(buf[pos:pos + 4], "little")
      pos += 4
    if off == 0 or off > len(out):
      raise ValueError("corrupt snappy block")
    for _ in range(ln):
      out.append(out[-off])
  if len(out) != n:
    raise ValueError("corrupt snappy block (length)")
  return bytes(out)


# ---- SSTable ------------------------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify):
  body, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
  if len(body) != size or len(trailer) != 5:
    raise ValueError("truncated table block")
  if verify and mask_crc(crc32c(body + trailer[:1])) != struct.unpack("<I", trailer[1:])[0]:
    raise ValueError("table block checksum mismatch")
  if trailer[0] == 1:
    body = _snappy_decompress(body)
  elif trailer[0] != 0:
    raise ValueError("unknown table block compression %d" % trailer[0])
  return body


def _block_entries(block):
  nrestart = struct.unpack("<I", block[-4:])[0]
  end = len(block) - 4 - 4 * nrestart
  pos, key, out = 0, b"", []
  while pos < end:
    shared, pos = _varint(block, pos)
    non_shared, pos = _varint(block, pos)
    vlen, pos = _varint(block, pos)
    key = key[:shared] + block[pos:pos + non_shared]
    pos += non_shared
    out.append((key, block[pos:pos + vlen]))
    pos += vlen
  return out


def read_table(path, verify=True):
  """[(key bytes, value bytes)] of an SSTable file, in key order."""
  buf = open(path, "rb").read()
  if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
    raise ValueError("%s is not a TensorFlow table file (bad magic)" % path)
  footer = buf[-48:]
  _, p = _varint(footer, 0)          # metaindex handle
  _, p = _varint(footer, p)
  ioff, p = _varint(footer, p)       # index handle
  isize, p = _varint(footer, p)
  out = []
  for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
    off, q = _varint(handle, 0)
    size, q = _varint(handle, q)
    out.extend(_block_entries(_read_block(buf, off, size, verify)))
  return out


class _BlockBuilder:
  def __init__(self, restart_interval=16):
    self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

  def add(self, key, value):
    shared = 0
    if self.count < self.ri:
      while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
        shared += 1
    else:
      self.restarts.append(len(self.buf))
      self.count = 0
    self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
    self.last = key
    self.count += 1

  def finish(self):
    return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=4096, restart_interval=16):
  """items: [(key bytes, value bytes)] sorted by key."""
  out = bytearray()

  def emit(block):
    off = len(out)
    out.extend(block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
    return _put_varint(off) + _put_varint(len(block))
  index = _BlockBuilder(1)
  bb, last = _BlockBuilder(restart_interval), None
  for key, value in items:
    if last is not None and key <= last:
      raise ValueError("table keys must be strictly increasing")
    bb.add(key, value)
    last = key
    if len(bb.buf) >= block_size:
      index.add(last, emit(bb.finish()))
      bb = _BlockBuilder(restart_interval)
  if bb.buf or not items:
    index.add(last if last is not None else b"", emit(bb.finish()))
  meta = emit(_BlockBuilder().finish())
  idx = emit(index.finish())
  footer = meta + idx
  out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
  with open(path, "wb") as f:
    f.write(bytes(out))


# ---- checkpoints ----------------------------------------------------------------------------------------------------------------
def _bf16_to_f32(raw):
  return (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)


def latest_checkpoint(directory):
  """tf.train.latest_checkpoint: the prefix named by <directory>/checkpoint, or None."""
  state = os.path.join(directory, "checkpoint")
  if not os.path.exists(state):
    return None
  m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', open(state).read(), re.M)
  if not m:
    return None
  p = m.group(1)
  return p if os.path.isabs(p) else os.path.join(directory, p)


class CheckpointReader:
  """tf.train.load_checkpoint(prefix) for V2 bundles: get_variable_to_shape_map(), has_tensor(), get_tensor()."""

  def __init__(self, prefix, verify=True):
    self.prefix = prefix
    index = prefix + ".index"
    if not os.path.exists(index):
      raise FileNotFoundError("no TensorFlow V2 checkpoint at %s (missing %s)" % (prefix, index))
    self.entries, self.num_shards, self.verify = {}, 1, verify
    for key, value in read_table(index, verify):
      if key == b"":
        for fn, wt, v in _fields(value):
          if fn == 1 and wt == 0:
            self.num_shards = v
          elif fn == 2 and wt == 0 and v == 1:
            raise ValueError("big-endian checkpoint")
      else:
        self.entries[key.decode()] = _parse_entry(value)
    self._shards = {}

  def get_variable_to_shape_map(self):
    return {k: list(e["shape"]) for k, e in self.entries.items()}

  def has_tensor(self, name):
    return name in self.entries

  def _shard(self, i):
    if i not in self._shards:
      self._shards[i] = np.memmap("%s.data-%05d-of-%05d" % (self.prefix, i, self.num_shards), dtype=np.uint8, mode="r")
    return self._shards[i]

  def get_tensor(self, name):
    e = self.entries[name]
    if e["sliced"]:
      raise NotImplementedError("%s is stored as slices (partitioned variable)" % name)
    raw = np.asarray(self._shard(e["shard_id"])[e["offset"]:e["offset"] + e["size"]]).tobytes()
    if len(raw) != e["size"]:
      raise ValueError("checkpoint data file is shorter than the index says (%s)" % name)
    if self.verify and e["crc32c"] is not None and mask_crc(crc32c_fast(raw)) != e["crc32c"]:
      raise ValueError("tensor checksum mismatch: %s" % name)
    if e["dtype"] == _DT_BFLOAT16:
      a = _bf16_to_f32(raw)
    elif e["dtype"] in _DTYPES:
      a = np.frombuffer(raw, dtype=np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<"))
    else:
      raise NotImplementedError("%s: TensorFlow dtype %d" % (name, e["dtype"]))
    return a.reshape(e["shape"]).copy()


def _read_v1_meta(value):
  """SavedTensorSliceMeta {tensor=1 repeated SavedSliceMeta {name=1, shape=2 TensorShapeProto, type=3, slice=4}} -> {name: (shape, dtype)}."""
  meta = {}
  for fn, wt, v in _fields(value):
    if fn != 1 or wt != 2:
      continue
    name, shape, dtype = None, (), 1
    for f2, w2, v2 in _fields(v):
      if f2 == 1 and w2 == 2:
        name = v2.decode()
      elif f2 == 2 and w2 == 2:
        shape = _parse_shape(v2)
      elif f2 == 3 and w2 == 0:
        dtype = v2
    if name is not None:
      meta[name] = (tuple(shape), dtype)
  return meta


def _read_v1(path):
  """A V1 checkpoint file: table values are SavedTensorSlices {meta=1, data=2 {name=1, slice=2, data=3 TensorProto}}.
  TensorSliceWriter fills ONLY the typed *_val field of a slice's TensorProto (tensor_slice_writer.h Fill<T>); shape and dtype
  of a tensor live in the SavedTensorSliceMeta stored under the empty key.  A TensorProto that does carry dtype / shape (other
  writers) is honoured."""
  out = {}
  records = list(read_table(path, verify=False))
  meta = {}
  for key, value in records:
    for fn, wt, v in _fields(value):
      if fn == 1 and wt == 2:
        meta.update(_read_v1_meta(v))
  for key, value in records:
    for fn, wt, v in _fields(value):
      if fn != 2 or wt != 2:
        continue
      name, tensor = None, None
      for f2, w2, v2 in _fields(v):
        if f2 == 1 and w2 == 2:
          name = v2.decode()
        elif f2 == 3 and w2 == 2:
          tensor = v2
      if name is None or tensor is None:
        continue
      dtype, shape, content, floats, ints, int64s = None, None, None, [], [], []
      for f3, w3, v3 in _fields(tensor):
        if f3 == 1 and w3 == 0:
          dtype = v3
        elif f3 == 2 and w3 == 2:
          shape = _parse_shape(v3)
        elif f3 == 4 and w3 == 2:
          content = v3
        elif f3 == 5:
          floats.append(np.frombuffer(v3, "<f4") if w3 == 2 else np.frombuffer(v3, "<f4"))
        elif f3 == 7:
          if w3 == 2:
            p, vals = 0, []
            while p < len(v3):
              x, p = _varint(v3, p)
              vals.append(_signed(x))
            ints.append(np.asarray(vals, np.int32))
          else:
            ints.append(np.asarray([_signed(v3)], np.int32))
        elif f3 == 10:
          if w3 == 2:
            p, vals = 0, []
            while p < len(v3):
              x, p = _varint(v3, p)
              vals.append(_signed(x))
            int64s.append(np.asarray(vals, np.int64))
          else:
            int64s.append(np.asarray([_signed(v3)], np.int64))
      if dtype is None:
        dtype = meta.get(name, ((), 1))[1]
      if shape is None:
        shape = meta.get(name, ((), 1))[0]
      if content is not None and dtype in _DTYPES:
        a = np.frombuffer(content, dtype=np.dtype(_DTYPES[dtype]).newbyteorder("<"))
      elif floats:
        a = np.concatenate(floats)
      elif ints:
        a = np.concatenate(ints)
      elif int64s:
        a = np.concatenate(int64s)
      else:
        continue
      n = int(np.prod(shape)) if shape else 1
      if a.size != n:
        if name in out:      # a variable saved in several slices: not produced by the checkpoints this path loads
          raise NotImplementedError("%s is stored in more than one slice" % name)
        if a.size == 1:
          a = np.full(n, a[0])
        else:
          raise ValueError("%s: %d values for shape %r" % (name, a.size, shape))
      out[name] = a.reshape(shape).copy()
  return out


def read_checkpoint(prefix, names=None, verify=True):
  """{variable name: numpy array} of a TensorFlow checkpoint: a V2 bundle prefix ('ckpt_pixrefer/pixrefernet-20000'),
  a directory holding a `checkpoint` state file, or a V1 single-file checkpoint ('allmodels/vgg_16.ckpt')."""
  if os.path.isdir(prefix):
    p = latest_checkpoint(prefix)
    if p is None:
      raise FileNotFoundError("no `checkpoint` state file in %s" % prefix)
    prefix = p
  if os.path.exists(prefix + ".index"):
    r = CheckpointReader(prefix, verify)
    keep = r.entries if names is None else [n for n in names if n in r.entries]
    return {n: r.get_tensor(n) for n in keep}
  if os.path.isfile(prefix):
    d = _read_v1(prefix)
    return d if names is None else {n: d[n] for n in names if n in d}
  raise FileNotFoundError("no TensorFlow checkpoint at %s" % prefix)


def is_tf_checkpoint(path):
  if not path:
    return False
  if os.path.exists(path + ".index"):
    return True
  if os.path.isfile(path) and os.path.getsize(path) >= 48:
    with open(path, "rb") as f:
      f.seek(-8, 2)
      return struct.unpack("<Q", f.read(8))[0] == TABLE_MAGIC
  return False


def write_checkpoint(prefix, tensors, update_state=True):
  """tf.train.Saver().save(sess, prefix) for {name: array}: <prefix>.index + <prefix>.data-00000-of-00001 (+ the `checkpoint`
  state file of the directory), readable by tf.train.load_checkpoint / Saver.restore.
  Every file is written under a temporary name and renamed into place (os.replace), the state file last: a process stopped in the
  middle of a save (torchrun ends every rank when one dies) leaves the previous checkpoint and its state file intact, never a
  truncated one that a restart would trust."""
  d = os.path.dirname(prefix)
  if d:
    os.makedirs(d, exist_ok=True)
  items, offset = [], 0
  tmp = ".tmp-%d" % os.getpid()
  data, index = prefix + ".data-00000-of-00001", prefix + ".index"
  with open(data + tmp, "wb") as f:
    for name in sorted(tensors, key=lambda s: s.encode()):
      a = np.ascontiguousarray(tensors[name])
      if a.dtype not in _DT_OF:
        raise TypeError("%s: dtype %s has no TensorFlow counterpart here" % (name, a.dtype))
      raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
      f.write(raw)
      crc = mask_crc(crc32c_fast(raw))
      items.append((name.encode(), _entry_proto(_DT_OF[a.dtype], a.shape, 0, offset, len(raw), crc)))
      offset += len(raw)
  header = b"\x08\x01" + b"\x1a\x02\x08\x01"      # num_shards = 1, (endianness LITTLE = default 0 omitted), version {producer = 1}
  write_table(index + tmp, [(b"", header)] + items)
  os.replace(data + tmp, data)
  os.replace(index + tmp, index)
  if update_state and d:
    update_checkpoint_state(d, prefix)
  return prefix


def update_checkpoint_state(directory, prefix, all_prefixes=None):
  """The `checkpoint` state file of a directory (tf.train.update_checkpoint_state), replaced atomically."""
  base = os.path.basename(prefix)
  lines = ['model_checkpoint_path: "%s"' % base]
  for p in (all_prefixes or [prefix]):
    lines.append('all_model_checkpoint_paths: "%s"' % os.path.basename(p))
  state = os.path.join(directory, "checkpoint")
  tmp = state + ".tmp-%d" % os.getpid()
  with open(tmp, "w") as f:
    f.write("\n".join(lines) + "\n")
  os.replace(tmp, state)


def list_checkpoints(directory, name):
  """Prefixes <directory>/<name>-<step> that have both files of a V2 bundle, oldest step first."""
  out = []
  if not os.path.isdir(directory):
    return out
  for fn in os.listdir(directory):
    m = re.match(r"^%s-(\d+)\.index$" % re.escape(name), fn)
    if m and os.path.exists(os.path.join(directory, fn[:-len(".index")] + ".data-00000-of-00001")):
      out.append((int(m.group(1)), os.path.join(directory, fn[:-len(".index")])))
  return [p for _, p in sorted(out)]


# ---- CRC32C of large buffers: lanes in lockstep (numpy) + GF(2) combination (the zlib crc32_combine construction) ------------
def _gf2_apply(mat, vec):
  """mat: 32 column images (uint32); vec: array of uint32 -> mat . vec over GF(2), element-wise over the array."""
  vec = np.asarray(vec, np.uint32)
  out = np.zeros_like(vec)
  for i in range(32):
    out ^= np.where((vec >> np.uint32(i)) & np.uint32(1), np.uint32(mat[i]), np.uint32(0)).astype(np.uint32)
  return out


def _gf2_square(mat):
  return [int(x) for x in _gf2_apply(mat, np.asarray(mat, np.uint32))]


def _zero_bytes_operator(nbytes):
  """Matrix that advances a (finalised) CRC32C over `nbytes` zero bytes."""
  one_bit = [0x82F63B78] + [1 << (i - 1) for i in range(1, 32)]
  op = _gf2_square(_gf2_square(_gf2_square(one_bit)))      # 8 zero bits = one zero byte
  result, n = None, nbytes
  while n:
    if n & 1:
      result = op if result is None else [int(x) for x in _gf2_apply(op, np.asarray(result, np.uint32))]
    n >>= 1
    if n:
      op = _gf2_square(op)
  return result


_NATIVE_CRC = None


def _native_crc():
  """vp_crc32c of libvp_hip.so (the SSE4.2 instruction, host code) when the library can be loaded; False otherwise."""
  global _NATIVE_CRC
  if _NATIVE_CRC is None:
    try:
      from .. import _lib
      _NATIVE_CRC = _lib.lib().vp_crc32c
    except Exception:
      _NATIVE_CRC = False
  return _NATIVE_CRC


def crc32c_fast(raw, lanes=16384):
  """crc32c(raw) for large buffers.  Through the library's hardware-instruction helper when it is loadable; otherwise in numpy: the
  buffer is cut into `lanes` equal pieces whose CRCs advance in lockstep through the byte table (one numpy step per byte
  position), then the piece CRCs fold pairwise: crc(A || B) = shift(crc(A), len(B)) ^ crc(B)."""
  n = len(raw)
  if n < (1 << 16):
    return crc32c(raw)
  f = _native_crc()
  if f:
    import ctypes
    buf = np.frombuffer(raw, np.uint8)
    return int(f(ctypes.c_void_p(buf.ctypes.data), n, 0))
  m = n // lanes
  body = np.frombuffer(raw, np.uint8, lanes * m).reshape(lanes, m).T.copy()      # [byte position][lane]
  tab = np.asarray(_CRC, np.uint32)
  c = np.full(lanes, 0xFFFFFFFF, np.uint32)
  for j in range(m):
    c = tab[(c ^ body[j]) & np.uint32(0xFF)] ^ (c >> np.uint32(8))
  c ^= np.uint32(0xFFFFFFFF)
  op, seg = _zero_bytes_operator(m), m
  while c.size > 1:
    c = _gf2_apply(op, c[0::2]) ^ c[1::2]
    seg *= 2
    if c.size > 1:
      op = _gf2_square(op)
  return crc32c(bytes(memoryview(raw)[lanes * m:]), int(c[0]))


def adam_steps_from_beta_powers(beta1_power, beta2_power, beta1, beta2):
  """Number of updates an AdamOptimizer has applied, from its saved non-slot variables: TF keeps beta^(t+1) after t updates
  (initial value beta, multiplied once per apply_gradients).  beta2_power is the better-conditioned one (0.999^t stays in
  float32 range for ~100k steps); 0 -> treated as 'very many' (the bias correction is 1 by then)."""
  for p, b in ((beta2_power, beta2), (beta1_power, beta1)):
    p = float(np.asarray(p).reshape(-1)[0])     # (a 0-d or 1-element array: explicit, NumPy deprecates the implicit form)
    if 0.0 < p < 1.0 and 0.0 < b < 1.0:
      return max(0, int(round(np.log(p) / np.log(b))) - 1)
  return 1000000
