"""File loaders (reference: generator/loader.py:9-119) without cv2 / librosa.

ImageLoader keeps the reference convention (cv2.imread: BGR, float32 in [0,1]); WavLoader returns mono
float32 in [-1,1] at the requested rate (reference: librosa.load).  Only PCM .wav input is supported
here: there is no audio decoder in the image (README of the reference feeds .aac through librosa/ffmpeg).
"""
import os

import numpy as np


class Loader(object):
  def __init__(self, root_path=None):
    self.root_path = root_path

  def _path(self, file_path):
    return os.path.join(self.root_path, file_path) if self.root_path else file_path


class ImageLoader(Loader):
  def __init__(self, root_path=None, resize=None):
    self.resize = resize
    Loader.__init__(self, root_path)

  def get_data(self, file_path):
    from PIL import Image
    img = Image.open(self._path(file_path)).convert("RGB")
    if self.resize is not None:
      img = img.resize((self.resize[0], self.resize[1]), Image.BILINEAR)
    data = np.asarray(img, dtype=np.float32)[:, :, ::-1]      # RGB -> BGR, as cv2.imread returns it
    return np.ascontiguousarray(data) / 255.0


class WavLoader(Loader):
  def __init__(self, root_path=None, sr=16000):
    self.sr = sr
    Loader.__init__(self, root_path)

  def get_data(self, file_path):
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    rate, data = wavfile.read(self._path(file_path))
    if data.dtype.kind == "i":
      data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
      data = (data.astype(np.float32) - 128.0) / 128.0
    else:
      data = data.astype(np.float32)
    if data.ndim > 1:
      data = data.mean(axis=1)                                  # librosa.load(mono=True)
    if rate != self.sr:
      g = np.gcd(int(rate), int(self.sr))
      data = resample_poly(data, self.sr // g, rate // g).astype(np.float32)
    return data


def _text_rows(path):
  with open(path) as f:
    return np.array([[float(v) for v in line.strip().split(",")] for line in f if line.strip()], dtype=np.float32)


class LandmarkLoader(Loader):
  """landmark.txt: one frame per line, 2x106 comma-separated coordinates, divided by norm_size (loader.py:58-66)."""

  def __init__(self, root_path=None, norm_size=128):
    Loader.__init__(self, root_path)
    self.norm_size = norm_size

  def get_data(self, file_path):
    path = self._path(file_path)
    return _text_rows(path) / self.norm_size if os.path.exists(path) else None


class BFMCoeffLoader(Loader):
  def get_data(self, file_path):
    with open(self._path(file_path)) as f:
      return np.array([[float(v) for v in line.strip().split(",")] for line in f if line.strip()], dtype=np.float32)
