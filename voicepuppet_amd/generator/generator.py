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
from ..runtime import Dataset
from .loader import ImageLoader

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
