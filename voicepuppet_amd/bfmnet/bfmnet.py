"""BFMNet inference with the reference's class surface (voicepuppet/bfmnet/bfmnet.py:125-333).
Training of BFMNet (vertex-space loss over the external BFM bases) is out of scope."""
import logging
import math
import os

import numpy as np

from ..builder import ModelBuilder
from ..config.configure import YParams
from ..runtime import Constant, Node, Placeholder

logger = logging.getLogger(__name__)


class BFMNet(ModelBuilder):

  def __init__(self, config_path):
    if (not os.path.exists(config_path)):
      logger.error('config_path not exists.')
      exit(0)
    self.__params = BFMNet.default_hparams(config_path)
    self.engine = None

  @staticmethod
  def default_hparams(config_path, name='default'):
    params = YParams(config_path, name)
    params.add_hparam('thinresnet_scale', [1, 32])
    params.add_hparam('thinresnet_output_channels', 256)
    params.add_hparam('encode_embedding_size', 256)
    params.add_hparam('rnn_hidden_size', 256)
    params.add_hparam('rnn_layers', 1)
    params.add_hparam('bfm_coeff_size', 64)
    params.training['learning_rate'] = 0.0001
    params.training['decay_steps'] = 10000
    params.training['decay_rate'] = 1
    return params

  @property
  def params(self):
    return self.__params

  def set_params(self, params):
    self.model_dir = params.model_dir
    self.batch_size = params.batch_size
    self.num_mel_bins = params.mel['num_mel_bins']
    self.frame_mfcc_scale = params.mel['sample_rate'] / params.frame_rate / params.mel['hop_step']
    assert (self.frame_mfcc_scale - int(self.frame_mfcc_scale) == 0), "sample_rate/hop_step must divided by frame_rate."
    self.thinresnet_pooling_size = [int(math.ceil(float(self.frame_mfcc_scale) / params.thinresnet_scale[0])),
                                    int(math.ceil(float(self.num_mel_bins) / params.thinresnet_scale[1]))]
    for k, want in (('thinresnet_output_channels', 256), ('encode_embedding_size', 256), ('rnn_hidden_size', 256),
                    ('rnn_layers', 1), ('bfm_coeff_size', 64)):
      if getattr(params, k) != want:
        raise NotImplementedError('%s=%r: the HIP executor is built for the reference defaults (%r)' % (k, getattr(params, k), want))

  def build_inference_op(self, ears, mfccs, seq_len):
    """ears [B,T,1], mfccs [B,5T,80] (node, tensor or array), seq_len [B] -> nodes with BFMCoeffDecoder [B,T,64]."""
    from ..audio import BFMNetEngine
    self._feeds = {'Ears': ears, 'Mfccs': mfccs, 'Seq_len': seq_len}
    shape = getattr(mfccs, 'shape', None)
    frames = int(shape[1]) // int(self.frame_mfcc_scale)
    self.engine = BFMNetEngine(self.batch_size, frames, self.num_mel_bins)
    return {k: Node(self, k) for k in ('Ears', 'Mfccs', 'MfccEncoder', 'RNNModule', 'BFMCoeffDecoder')}

  def load_params(self, params):
    self.engine.load_params(params)

  def init_variables(self, seed=None):
    """Random stand-in weights (xavier-like kernels, unit moving variance) when no checkpoint is available."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, _, shape in self.engine.manifest:
      if name.endswith('moving_variance'):
        p[name] = np.ones(shape, np.float32)
      elif name.endswith('gates/bias'):
        p[name] = np.ones(shape, np.float32)
      elif name.endswith(('moving_mean', 'beta', 'bias')):
        p[name] = np.zeros(shape, np.float32)
      else:
        fan_in = shape[0] * shape[1] if 'depthwise' in name else int(np.prod(shape[:-1]))
        p[name] = rng.normal(0, np.sqrt(2.0 / max(fan_in, 1)), shape).astype(np.float32)
    self.engine.load_params(p)

  def restore(self, path):
    """tf.train.Saver().restore(sess, 'ckpt_bfmnet/bfmnet-65000') (infer_bfmnet.py:192, infer_bfmvid.py:217): a TensorFlow
    checkpoint prefix / directory, or an .npz keyed by the same variable names."""
    if path.endswith('.npz'):
      z = np.load(path)
      d = {k: z[k] for k in z.files}
    else:
      from ..utils import tf_checkpoint
      d = tf_checkpoint.read_checkpoint(path, names=[n for n, _, _ in self.engine.manifest])
    missing = [n for n, _, _ in self.engine.manifest if n not in d]
    if missing:
      raise KeyError('%s holds no %s (%d BFMNet variables missing)' % (path, missing[0], len(missing)))
    self.engine.load_params(d)

  def execute(self, names, feed_dict):
    import torch
    vals = {}
    for key, node in self._feeds.items():
      v = feed_dict[node] if isinstance(node, Placeholder) else (node.value if isinstance(node, Constant) else node)
      vals[key] = v
    t = lambda v: (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v, dtype=np.float32))).to('cuda', torch.float32)
    seq = vals['Seq_len']
    seq = seq.cpu().numpy() if torch.is_tensor(seq) else np.asarray(seq)
    coeff = self.engine.forward(t(vals['Ears']), t(vals['Mfccs']), seq.astype(np.int32))
    out = {}
    for n in names:
      if n == 'BFMCoeffDecoder':
        out[n] = coeff.cpu().numpy()
      elif n in ('MfccEncoder', 'RNNModule'):
        out[n] = self.engine.tensor(n).cpu().numpy()
      else:
        out[n] = t(vals[n]).cpu().numpy()
    return out
