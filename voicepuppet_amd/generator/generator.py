"""Data generators on the hot path (reference: generator/generator.py:23-114, 924-1040).

DataGenerator.extract_mfcc runs the fused log-mel HIP path (it is NOT a cepstrum: SURVEY.md fact 1);
PixReferDataGenerator reproduces the input layout of the training step from 1536x512 jpg triptychs
(frame | 3-D face | matte) and falls back to synthetic batches of the same layout when asked to.
"""
import logging
import math
import os
import random

import numpy as np

from ..config.configure import YParams
from ..runtime import Dataset, DatasetIterator, IteratorNext
from .loader import BFMCoeffLoader, ImageLoader, LandmarkLoader, WavLoader

logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
logger = logging.getLogger(__name__)


class DataGenerator(object):
  def __init__(self, config_path):
    if (not os.path.exists(config_path)):
      logger.error('config_path not exists.')
      exit(0)
    self._params = type(self).default_hparams(config_path)
    self._logmel = {}

  @staticmethod
  def default_hparams(config_path, name='default'):
    return YParams(config_path, name)

  @property
  def params(self):
    return self._params

  def set_params(self, params):
    self.sample_rate = params.mel['sample_rate']
    self.num_mel_bins = params.mel['num_mel_bins']
    self.win_length = params.mel['win_length']
    self.hop_step = params.mel['hop_step']
    self.fft_length = params.mel['fft_length']
    self.frame_rate = params.frame_rate
    self.frame_wav_scale = self.sample_rate / self.frame_rate
    self.frame_mfcc_scale = self.frame_wav_scale / self.hop_step
    assert (self.frame_mfcc_scale - int(self.frame_mfcc_scale) == 0), "sample_rate/hop_step must divided by frame_rate."
    self.frame_mfcc_scale = int(self.frame_mfcc_scale)

  def iterator(self):
    raise NotImplementedError('iterator not implemented.')

  def get_dataset(self):
    raise NotImplementedError('get_dataset not implemented.')

  def extract_mfcc(self, pcm):
    """[batch, samples] mono PCM in [-1,1] -> log-mel [batch, 1+(samples-win)//hop, num_mel_bins] (device tensor)."""
    import torch
    from ..audio import LogMel
    pcm = torch.as_tensor(np.asarray(pcm, dtype=np.float32) if not torch.is_tensor(pcm) else pcm, dtype=torch.float32).cuda()
    key = tuple(pcm.shape)
    if key not in self._logmel:
      self._logmel[key] = LogMel(pcm.shape[0], pcm.shape[1], self.sample_rate, self.num_mel_bins, self.win_length,
                                 self.hop_step, self.fft_length, 80.0, 7600.0)
    return self._logmel[key](pcm)

  def ear_compute(self, landmarks):
    ears = []
    for ps in landmarks:
      ps = [float(x) for x in ps]
      d = lambda a, b: math.sqrt((ps[a] - ps[b]) ** 2 + (ps[a + 1] - ps[b + 1]) ** 2)
      ear1 = (d(74, 82) + d(76, 80)) / d(72, 78)
      ear2 = (d(86, 94) + d(88, 92)) / d(84, 90)
      ears.append([(ear1 + ear2) / 2])
    return np.array(ears)

  def split_bfmcoeff(self, coeff):
    return coeff[:80], coeff[80:144], coeff[144:224], coeff[224:227], coeff[227:254], coeff[254:]

  def pose_compute(self, bfmcoeffs):
    return np.array([self.split_bfmcoeff(c)[3] for c in bfmcoeffs])


def first_nonsilent_sample(pcm, top_db=20, frame_length=2048, hop_length=512):
  """Start of the first interval librosa.effects.split(pcm, top_db) returns (generator.py:457-459).  librosa is a third-party
  dependency the reference does not pin; this restates the published algorithm of the 0.7/0.8 releases contemporary with it: frame-wise
  mean square over centred frames (reflect padding), in dB relative to the loudest frame (amin 1e-10); a frame is non-silent above
  -top_db; frame index -> sample by * hop_length, clipped to the signal length."""
  y = np.asarray(pcm, dtype=np.float32)
  pad = frame_length // 2
  yp = np.pad(y, (pad, pad), mode='reflect' if y.shape[0] > pad else 'constant')
  n = 1 + (yp.shape[0] - frame_length) // hop_length
  if n <= 0:
    return 0
  sq = np.concatenate([[0.0], np.cumsum(yp.astype(np.float64) ** 2)])
  idx = np.arange(n) * hop_length
  mse = (sq[idx + frame_length] - sq[idx]) / frame_length
  amin = 1e-10
  db = 10.0 * np.log10(np.maximum(amin, mse)) - 10.0 * np.log10(np.maximum(amin, mse.max()))
  loud = np.flatnonzero(db > -top_db)
  if loud.size == 0:
    return 0
  return int(min(loud[0] * hop_length, y.shape[0]))


class BFMNetDataGenerator(DataGenerator):
  """(bfmcoeff [T,257], ear [T,1], pcm, T) slices of 24 frames per clip folder, batched and turned into log-mel features on the
  device (generator.py:377-500).  The list format is the one of makelist: `folder|frame count` per line."""

  SLICE = 24   # generator.py:455 (rnd_len)

  def __init__(self, config_path):
    if (not os.path.exists(config_path)):
      logger.error('config_path not exists.')
      exit(0)
    self._params = BFMNetDataGenerator.default_hparams(config_path)
    self._logmel = {}

  @staticmethod
  def default_hparams(config_path, name='default'):
    params = YParams(config_path, name)
    params.add_hparam('dataset_path', params.train_dataset_path)
    params.add_hparam('max_squence_len', 30)
    params.add_hparam('min_squence_len', 20)
    params.add_hparam('shuffle_bufsize', 1000)
    params.add_hparam('batch_size', 8)
    return params

  def set_params(self, params):
    DataGenerator.set_params(self, params)
    amd = params.get('amd') or {}
    self.synthetic = amd.get('synthetic_data', 'auto')
    if os.path.exists(params.dataset_path):
      self.data_list = open(params.dataset_path).readlines()
    elif self.synthetic in ('auto', True, 'true', 'yes'):
      logger.warning('%s not found: using synthetic BFMNet clips', params.dataset_path)
      self.data_list = None
    else:
      raise IOError('dataset list not found: %s' % params.dataset_path)
    self.shuffle_bufsize = params.shuffle_bufsize
    self.landmark_name = params.sample_file['landmark_name']
    self.wav_name = params.sample_file['wav_name']
    self.bfmcoeff_name = params.sample_file['bfmcoeff_name']
    self.max_squence_len = params.max_squence_len
    self.min_squence_len = params.min_squence_len
    self.batch_size = params.batch_size

  def pcm_length(self, frames):
    """Samples that give exactly frames * frame_mfcc_scale log-mel rows (generator.py:476)."""
    return self.hop_step * (frames * self.frame_mfcc_scale - 1) + self.win_length

  def slices(self, bfmcoeffs, ear, pcm):
    """The slicing of one clip (generator.py:455-481): drop the leading silence, replace the identity coefficients by the clip mean,
    cut consecutive 24-frame slices with the PCM window that keeps log-mel rows and frames aligned."""
    rnd_len = self.SLICE
    start = first_nonsilent_sample(pcm, top_db=20)
    sil_rm_start = int(start // self.frame_wav_scale)
    pcm = pcm[start:]
    bfmcoeffs = np.array(bfmcoeffs[sil_rm_start:, :], dtype=np.float32)
    if bfmcoeffs.shape[0] == 0:
      return
    bfmcoeffs[:, :80] = np.mean(bfmcoeffs[:, :80], 0, keepdims=True)
    # the reference slices `ear` from the un-trimmed start while the coefficients are trimmed (generator.py:471-472)
    for i in range(bfmcoeffs.shape[0] // rnd_len):
      bfmcoeff_slice = bfmcoeffs[i * rnd_len: (i + 1) * rnd_len, :]
      ear_slice = ear[i * rnd_len: (i + 1) * rnd_len, :]
      pcm_start = int(i * rnd_len * self.frame_wav_scale)
      pcm_length = self.pcm_length(rnd_len)
      if (pcm.shape[0] < pcm_start + pcm_length):
        pcm = np.pad(pcm, (0, pcm_start + pcm_length - pcm.shape[0]), 'constant', constant_values=(0))
      yield bfmcoeff_slice, ear_slice.astype(np.float32), pcm[pcm_start: pcm_start + pcm_length].astype(np.float32), bfmcoeff_slice.shape[0]

  def _synthetic(self, rand=random):
    rng = np.random.default_rng(rand.randint(0, 2 ** 31))
    T = self.SLICE
    n = self.pcm_length(T)
    t = np.arange(n) / self.sample_rate
    while True:
      coeff = rng.normal(0, 0.5, (T, 257)).astype(np.float32)
      coeff[:, :80] = coeff[:1, :80]
      f0 = rng.uniform(90, 300)
      pcm = (0.3 * np.sin(2 * np.pi * f0 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.normal(size=n)).astype(np.float32)
      yield coeff, rng.uniform(0.6, 0.9, (T, 1)).astype(np.float32), pcm, T

  def iterator(self, rand=random):
    """rand: the source of the order-defining draws (shuffle of the file list, seed of the synthetic clips): the module-level `random`
    as in the reference, or the private random.Random of ONE iterator (_MfccIterator: its background thread must not share a generator
    with the main thread, nor with another iterator over the same DataGenerator)."""
    if self.data_list is None:
      for s in self._synthetic(rand):
        yield s
      return
    bfmcoeff_loader = BFMCoeffLoader()
    landmark_loader = LandmarkLoader(norm_size=1)
    wav_loader = WavLoader(sr=self.sample_rate)
    rand.shuffle(self.data_list)
    for line in self.data_list:
      folder, img_count = line.strip().split('|')
      img_count = int(img_count)
      paths = [os.path.join(folder, n) for n in (self.bfmcoeff_name, self.landmark_name, self.wav_name)]
      if img_count <= 0 or not all(os.path.exists(p) for p in paths):
        continue
      bfmcoeffs = bfmcoeff_loader.get_data(paths[0])
      landmark = landmark_loader.get_data(paths[1])
      pcm = wav_loader.get_data(paths[2])
      if (bfmcoeffs.shape[0] == img_count and landmark.shape[0] == img_count):
        ear = 1 - self.ear_compute(landmark)
        for s in self.slices(bfmcoeffs, ear, pcm):
          yield s

  def process_data(self, bfmcoeff, ear, pcm, seq_len):
    return bfmcoeff, ear, self.extract_mfcc(pcm), seq_len

  def get_dataset(self):
    """Batches of (bfmcoeff [B,24,257], ear [B,24,1], mfcc [B,120,80] (device), seq_len [B]); every slice has the same length, so the
    reference's padded_batch pads nothing."""
    self.set_params(self._params)
    T = self.SLICE
    return _MfccDataset(self, self.iterator, ([T, 257], [T, 1], [self.pcm_length(T)], []), self.batch_size, self.shuffle_bufsize)


class _MfccIterator(DatasetIterator):
  def __init__(self, ds):
    DatasetIterator.__init__(self, ds)
    # The worker thread's order-defining draws come from a generator of THIS iterator, seeded from the module-level state where the
    # iterator is made (on the caller's thread) WITHOUT advancing it: the sample order is a function of random.seed() alone, whatever
    # the main thread or another iterator over the same DataGenerator draws meanwhile, and a caller that seeded `random` keeps its stream
    # (the seed is DRAWN from the stream and the stream put back: hash(random.getstate()) - round 5 - is not a function of the seed, the state
    # tuple ends in None and hash(None) is the object's address before CPython 3.12, so two processes after random.seed(5) disagreed: ADVICE r5)
    st = random.getstate()
    self._rand = random.Random(random.getrandbits(64))
    random.setstate(st)

  def _samples(self):
    while True:   # repeat()
      n = 0
      for s in self.ds.owner.iterator(self._rand):
        n += 1
        yield s
      if n == 0:
        raise RuntimeError("the dataset generator yielded nothing")

  def get_next(self):
    g, b = self.ds.owner, self.ds.batch_size
    T = g.SLICE
    shapes = ((b, T, 257), (b, T, 1), (b, T * g.frame_mfcc_scale, g.num_mel_bins), (b,))
    return tuple(IteratorNext(self, k, s) for k, s in enumerate(shapes))

  def _host_batches(self):
    """The host half of a batch (file reads / slicing / shuffling / stacking: 19 ms for 32 synthetic clips) on a background thread, two
    batches ahead: the training loop reads its loss after every step (train_bfmnet.py:96-97 prints it), and while it waits for the
    device the interpreter is free to prepare the next batch.  The device half (log-mel) stays on the caller's thread."""
    import queue
    import threading
    import weakref
    q = queue.Queue(maxsize=2)
    stop = threading.Event()
    END = object()
    nxt = DatasetIterator.next_batch
    me = weakref.ref(self)                 # the thread must not keep the iterator alive: dropping the iterator stops the thread

    def put(item):
      while not stop.is_set():
        try:
          q.put(item, timeout=0.2)
          return True
        except queue.Full:
          if me() is None:
            return False
      return False

    def work():
      try:
        while not stop.is_set():
          it = me()
          if it is None:
            return
          batch = nxt(it)
          del it
          if not put(batch):
            return
      except StopIteration:               # a finite source ran out: a sentinel, not an exception object (PEP 479)
        put(END)
      except BaseException as e:          # hand the failure to the consumer
        put(e)
    th = threading.Thread(target=work, daemon=True, name="bfmnet-batches")
    self._hb_stop, self._hb_thread = stop, th
    th.start()
    try:
      while True:
        item = q.get()
        if item is END:
          return
        if isinstance(item, BaseException):
          raise RuntimeError("the batch thread failed") from item
        yield item
    finally:
      stop.set()

  def close(self):
    """Stop the background batch thread (also happens when the iterator is garbage-collected)."""
    if getattr(self, "_hb_stop", None) is not None:
      self._hb_stop.set()
      self._hb_thread.join(timeout=2.0)
      self._hb = None

  def __del__(self):
    if getattr(self, "_hb_stop", None) is not None:
      self._hb_stop.set()

  def next_batch(self):
    if getattr(self, "_hb", None) is None:
      self._hb = self._host_batches()
    coeff, ear, pcm, n = next(self._hb)
    return self.ds.owner.process_data(coeff, ear, pcm, n.astype(np.int32))


class _MfccDataset(Dataset):
  """dataset.map(process_data): PCM -> log-mel on the device after batching (generator.py:483-500)."""

  def __init__(self, owner, *a, **kw):
    Dataset.__init__(self, *a, **kw)
    self.owner = owner

  def make_one_shot_iterator(self):
    return _MfccIterator(self)


def pack_sample(example_rgb3, img_rgb3, img_size):
  """The channel packing of generator.py:1006-1019.  Both arguments are [S, 3S, 3] float RGB triptychs
  (target | 3dface | mask) of the example frame and of the current frame."""
  imgs = np.array([example_rgb3, img_rgb3])
  inputs = imgs[:, :, img_size:img_size * 2, :].transpose((1, 2, 0, 3)).reshape([img_size, img_size, 6])
  targets = imgs[:, :, :img_size, :]
  masks = imgs[:, :, img_size * 2:, :]
  fg_inputs = (targets * masks).transpose([1, 2, 0, 3]).reshape([img_size, img_size, 6])
  return inputs, fg_inputs, targets[1, ...], masks[1, ...]


class PixReferDataGenerator(DataGenerator):
  def __init__(self, config_path):
    if (not os.path.exists(config_path)):
      logger.error('config_path not exists.')
      exit(0)
    self._params = PixReferDataGenerator.default_hparams(config_path)
    self._logmel = {}

  @staticmethod
  def default_hparams(config_path, name='default'):
    params = YParams(config_path, name)
    params.add_hparam('dataset_path', params.train_dataset_path)
    params.add_hparam('shuffle_bufsize', 100)
    params.add_hparam('batch_size', 2)
    params.add_hparam('img_size', int((params.get('amd') or {}).get('img_size', 512)))
    params.add_hparam('crop_ratio', 0.9)
    params.add_hparam('seq_len', 8)
    return params

  def set_params(self, params):
    amd = params.get('amd') or {}
    self.synthetic = amd.get('synthetic_data', 'auto')
    if os.path.exists(params.dataset_path):
      self.data_list = open(params.dataset_path).readlines()
    elif self.synthetic in ('auto', True, 'true', 'yes'):
      logger.warning('%s not found: using synthetic PixReferNet batches', params.dataset_path)
      self.data_list = None
    else:
      raise IOError('dataset list not found: %s' % params.dataset_path)
    self.shuffle_bufsize = params.shuffle_bufsize
    self.batch_size = params.batch_size
    self.img_size = params.img_size
    self.crop_ratio = params.crop_ratio
    self.seq_len = params.seq_len

  def _load_triptych(self, image_loader, path):
    """jpg (S x 3S BGR) -> random square crop + resize of the three panels -> [S, 3S, 3] RGB float."""
    from PIL import Image
    S = self.img_size
    rsize = random.randint(int(S * self.crop_ratio), S)
    rx = random.randint(0, S - rsize)
    ry = random.randint(0, S - rsize)
    img = image_loader.get_data(path)[:, :, ::-1]                 # cv2.cvtColor(BGR2RGB)
    img = np.concatenate([img[:, :S, :], img[:, S:S * 2, :], img[:, S * 2:, :]], axis=-1)
    img = img[rx:rsize + rx, ry:rsize + ry, :]
    planes = [np.asarray(Image.fromarray(np.ascontiguousarray(img[:, :, c]), mode="F").resize((S, S), Image.BILINEAR))
              for c in range(9)]                                   # cv2.resize(..., INTER_LINEAR)
    img = np.stack(planes, axis=-1)
    return np.concatenate([img[:, :, :3], img[:, :, 3:6], img[:, :, 6:]], axis=1)

  def _synthetic(self):
    S = self.img_size
    rng = np.random.default_rng(random.randint(0, 2 ** 31))
    yy, xx = np.mgrid[0:S, 0:S]
    while True:
      def trip():
        r = np.sqrt((yy - S / 2 - rng.normal(0, S / 40)) ** 2 + (xx - S / 2 - rng.normal(0, S / 40)) ** 2)
        mask = np.clip((0.35 * S + 4 - r) / 8, 0, 1)[..., None].repeat(3, 2)
        lo = rng.uniform(size=(S // 8, S // 8, 6)).astype(np.float32).repeat(8, 0).repeat(8, 1)
        return np.concatenate([lo[..., :3], lo[..., 3:] * mask, mask], axis=1).astype(np.float32)
      yield pack_sample(trip(), trip(), S)

  def iterator(self):
    if self.data_list is None:
      for s in self._synthetic():
        yield s
      return
    image_loader = ImageLoader()
    random.shuffle(self.data_list)
    for line in self.data_list:
      folder, img_count = line.strip().split('|')
      img_count = int(img_count)
      for i in range(img_count):
        rnd_idx = random.randint(0, img_count - 1)
        example_img = self._load_triptych(image_loader, os.path.join(folder, '{}.jpg'.format(rnd_idx)))
        img = self._load_triptych(image_loader, os.path.join(folder, '{}.jpg'.format(i)))
        yield pack_sample(example_img, img, self.img_size)

  def get_dataset(self):
    self.set_params(self._params)
    S = self.img_size
    return Dataset(self.iterator, ([S, S, 6], [S, S, 6], [S, S, 3], [S, S, 3]), self.batch_size, self.shuffle_bufsize)

  # ---- the same dataset with the per-sample arithmetic on the device ---------------------------------------------
  def _frame_samples(self):
    """(example frame, current frame, crops) per sample, as the device pipeline takes them: the two DECODED jpg triptychs
    [S, 3S, 3] uint8 BGR (what cv2.imread returns) and the (rx, ry, rsize) each would be cropped with (generator.py:975-977) - the
    crop / resize / packing themselves run in vp_pixrefer_pack_frames.  Decoding runs on a thread pool (PIL releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from .device_pipeline import draw_crop
    S = self.img_size

    def crops():
      return np.array([draw_crop(S, self.crop_ratio), draw_crop(S, self.crop_ratio)], np.int32)
    if self.data_list is None:
      # synthetic frames: a small pool of random triptychs (smooth 8 x 8 blocks, a disc matte), fresh crops every time
      rng = np.random.default_rng(random.randint(0, 2 ** 31))
      yy, xx = np.mgrid[0:S, 0:S]
      pool = []
      for _ in range(16):
        r = np.sqrt((yy - S / 2 - rng.normal(0, S / 40)) ** 2 + (xx - S / 2 - rng.normal(0, S / 40)) ** 2)
        mask = np.clip((0.35 * S + 4 - r) / 8, 0, 1)[..., None].repeat(3, 2)
        lo = rng.uniform(size=(S // 8, S // 8, 6)).astype(np.float32).repeat(8, 0).repeat(8, 1)
        pool.append(np.ascontiguousarray((np.concatenate([lo[..., :3], lo[..., 3:] * mask, mask], axis=1) * 255).astype(np.uint8)))
      while True:
        yield pool[int(rng.integers(16))], pool[int(rng.integers(16))], crops()

    def decode(path):
      return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)[:, :, ::-1])      # BGR, as cv2.imread
    workers = max(2, min(16, (os.cpu_count() or 4) - 1))
    with ThreadPoolExecutor(max_workers=workers) as pool:
      while True:
        random.shuffle(self.data_list)
        jobs = []
        for line in self.data_list:
          folder, img_count = line.strip().split('|')
          img_count = int(img_count)
          for i in range(img_count):
            rnd_idx = random.randint(0, img_count - 1)
            jobs.append((pool.submit(decode, os.path.join(folder, '{}.jpg'.format(rnd_idx))),
                         pool.submit(decode, os.path.join(folder, '{}.jpg'.format(i)))))
            while len(jobs) > 4 * workers:          # a bounded number of decodes in flight, handed out in order
              a, b = jobs.pop(0)
              yield a.result(), b.result(), crops()
        for a, b in jobs:
          yield a.result(), b.result(), crops()

  def get_device_dataset(self):
    """get_dataset() for a GPU training loop: same sample order semantics (shuffle buffer, repeat, batches), but a batch is four
    float32 DEVICE tensors produced by vp_pixrefer_pack_frames from uint8 frames, copied and packed on a side stream under the previous
    step (generator/device_pipeline.py)."""
    self.set_params(self._params)
    return _DeviceFrameDataset(self)


class _DeviceFrameIterator(object):
  def __init__(self, ds):
    self.ds = ds
    self._pf = None

  def get_next(self):
    b, S = self.ds.batch_size, self.ds.owner.img_size
    return tuple(IteratorNext(self, k, (b, S, S, c)) for k, c in enumerate((6, 6, 3, 3)))

  def _batches(self):
    """Batches as PINNED torch tensors the prefetcher copies from directly (no staging copy): a ring of four, since a batch must stay
    unchanged until two batches later (FramePrefetcher, which host-waits for the copies of batch k-2 before it asks for batch k: slot
    k % 4 was last read for batch k-4).  The samples of a batch are written straight into the ring slot."""
    import torch
    g = self.ds.owner
    N, S = self.ds.batch_size, g.img_size
    ring = [(torch.empty(N, S, 3 * S, 3, dtype=torch.uint8).pin_memory(), torch.empty(N, S, 3 * S, 3, dtype=torch.uint8).pin_memory(),
             torch.empty(N, 2, 3, dtype=torch.int32).pin_memory()) for _ in range(4)]
    views = [tuple(t.numpy() for t in slot) for slot in ring]
    buf, it = [], g._frame_samples()
    rng = np.random.default_rng()
    k = 0
    while True:
      ex, cur, crops = views[k % 4]
      for j in range(N):
        while len(buf) < max(1, g.shuffle_bufsize):
          buf.append(next(it))
        s = buf.pop(int(rng.integers(len(buf))) if g.shuffle_bufsize > 1 else 0)
        ex[j] = s[0]; cur[j] = s[1]; crops[j] = s[2]
      yield ring[k % 4]
      k += 1

  def next_batch(self):
    if self._pf is None:
      from .device_pipeline import FramePrefetcher
      self._pf = FramePrefetcher(self._batches(), self.ds.batch_size, self.ds.owner.img_size)
    return self._pf.next()


class _DeviceFrameDataset(object):
  def __init__(self, owner):
    self.owner, self.batch_size = owner, owner.batch_size

  def make_one_shot_iterator(self):
    return _DeviceFrameIterator(self)
