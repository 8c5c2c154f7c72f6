"""BFMNet with the reference's class surface (voicepuppet/bfmnet/bfmnet.py:125-333): build_inference_op runs the inference plan
(voicepuppet_amd.audio.BFMNetEngine), build_train_op / build_eval_op the training step (train_engine.BFMNetTrainEngine, SURVEY.md
8f-4).  The vertex-space loss needs the external face model (BFM/BFM_model_front.mat + mouth_idx.npy under params.model_dir,
bfmnet.py:133-137); when it is absent and `amd.synthetic_data` allows it a random stand-in of the same shapes is used."""
import logging
import math
import os

import numpy as np

from ..builder import ModelBuilder
from ..config.configure import YParams
from ..runtime import Constant, IteratorNext, Node, Placeholder

logger = logging.getLogger(__name__)


def random_variables(seed=None):
  """{variable name: array} of a freshly initialised BFMNet (not TensorFlow's initialiser stream: no checkpoint, no parity claim)."""
  from ..audio import bfmnet_manifest
  rng = np.random.default_rng(seed)
  p = {}
  for name, _, shape in bfmnet_manifest():
    if name.endswith('moving_variance'):
      p[name] = np.ones(shape, np.float32)
    elif name.endswith('gates/bias'):
      p[name] = np.ones(shape, np.float32)
    elif name.endswith(('moving_mean', 'beta', 'bias')):
      p[name] = np.zeros(shape, np.float32)
    else:
      fan_in = shape[0] * shape[1] if 'depthwise' in name else int(np.prod(shape[:-1]))
      p[name] = rng.normal(0, np.sqrt(2.0 / max(fan_in, 1)), shape).astype(np.float32)
  return p


class BFMNet(ModelBuilder):

  def __init__(self, config_path):
    if (not os.path.exists(config_path)):
      logger.error('config_path not exists.')
      exit(0)
    self.__params = BFMNet.default_hparams(config_path)
    self.engine = None
    self.train_engine = None
    self.global_step = 0

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
    self.learning_rate = params.training['learning_rate']
    self.max_grad_norm = params.training.get('max_grad_norm', 50)
    self.decay_steps = params.training['decay_steps']
    self.decay_rate = params.training['decay_rate']
    self.drop_rate = params.training.get('drop_rate', 0.25)
    self.synthetic = (params.get('amd') or {}).get('synthetic_data', 'auto')
    # opt-in (params.yml: amd: {reference_decoder_dropout: true}): the reference's BFMCoeffDecoder keeps its two tf.nn.dropout at
    # inference (bfmnet.py:114,116); off = the deterministic expectation
    self.reference_decoder_dropout = bool((params.get('amd') or {}).get('reference_decoder_dropout', False))
    for k, want in (('thinresnet_output_channels', 256), ('encode_embedding_size', 256), ('rnn_hidden_size', 256),
                    ('rnn_layers', 1), ('bfm_coeff_size', 64)):
      if getattr(params, k) != want:
        raise NotImplementedError('%s=%r: the HIP executor is built for the reference defaults (%r)' % (k, getattr(params, k), want))

  def _frames(self, mfccs):
    return int(getattr(mfccs, 'shape')[1]) // int(self.frame_mfcc_scale)

  def build_inference_op(self, ears, mfccs, seq_len):
    """ears [B,T,1], mfccs [B,5T,80] (node, tensor or array), seq_len [B] -> nodes with BFMCoeffDecoder [B,T,64]."""
    from ..audio import BFMNetEngine
    self._feeds = {'Ears': ears, 'Mfccs': mfccs, 'Seq_len': seq_len}
    self.engine = BFMNetEngine(self.batch_size, self._frames(mfccs), self.num_mel_bins)
    return {k: Node(self, k) for k in ('Ears', 'Mfccs', 'MfccEncoder', 'RNNModule', 'BFMCoeffDecoder')}

  # ---- training (bfmnet.py:215-323) -------------------------------------------------------------------------------------------
  def load_facemodel(self):
    """exBase [3n,64] and the mouth-weighted vertex mask (bfmnet.py:133-137); idBase / meanshape cancel in the loss."""
    mat = os.path.join(self.model_dir, 'BFM_model_front.mat')
    idx = os.path.join(self.model_dir, 'mouth_idx.npy')
    if os.path.exists(mat) and os.path.exists(idx):
      from scipy.io import loadmat
      ex = np.asarray(loadmat(mat)['exBase'], dtype=np.float32)
      vm = np.ones([ex.shape[0] // 3, 3], dtype=np.float32)
      vm[np.load(idx).reshape(-1)] = 10.0
      return {'exBase': ex, 'vmask': vm.reshape(-1)}
    if self.synthetic not in ('auto', True, 'true', 'yes'):
      raise IOError('face model not found: %s, %s' % (mat, idx))
    logger.warning('%s not found: using a random stand-in face model (35709 vertices)', mat)
    rng = np.random.default_rng(0)
    vm = np.ones([35709, 3], dtype=np.float32)
    vm[rng.choice(35709, 1800, replace=False)] = 10.0
    return {'exBase': rng.normal(0, 0.05, (35709 * 3, 64)).astype(np.float32), 'vmask': vm.reshape(-1)}

  def _train_engine(self, frames):
    from .train_engine import BFMNetTrainEngine
    if self.train_engine is None:
      self.train_engine = BFMNetTrainEngine(self.batch_size, frames, self.load_facemodel(), lr=self.learning_rate,
                                            max_grad_norm=self.max_grad_norm, num_mel_bins=self.num_mel_bins)
      self.init_variables(target=self.train_engine)
    elif self.train_engine.T != frames:
      raise ValueError('train and eval graphs share one engine: %d frames per clip, got %d' % (self.train_engine.T, frames))
    return self.train_engine

  def build_train_op(self, bfm_coeff_seq, ears, mfccs, seq_len):
    """bfm_coeff_seq [B,T,257], ears [B,T,1], mfccs [B,5T,80], seq_len [B].  Every clip of a batch is padded to the same T (the
    generator cuts fixed 24-frame slices, generator.py:455); the plan is built for that T."""
    self._train_feeds = {'BFM_coeff_seq': bfm_coeff_seq, 'Ears': ears, 'Mfccs': mfccs, 'Seq_len': seq_len}
    self._train_engine(self._frames(mfccs))
    keys = ('BFM_coeff_seq', 'Ears', 'Mfccs', 'Seq_len', 'BFMCoeffDecoder', 'Loss', 'Global_step', 'Lr', 'Train_op', 'Grads', 'Tvars')
    return self._nodes('train', keys)

  def build_eval_op(self, bfm_coeff_seq, ears, mfccs, seq_len):
    """Inference-mode forward (moving statistics, no dropout) of the variables being trained + the same cost (bfmnet.py:273-289)."""
    from ..audio import BFMNetEngine
    self._eval_feeds = {'BFM_coeff_seq': bfm_coeff_seq, 'Ears': ears, 'Mfccs': mfccs, 'Seq_len': seq_len}
    frames = self._frames(mfccs)
    self._train_engine(frames)
    self._eval_engine = BFMNetEngine(self.batch_size, frames, self.num_mel_bins)
    return self._nodes('eval', ('BFM_coeff_seq', 'Ears', 'Mfccs', 'Seq_len', 'MfccEncoder', 'RNNModule', 'BFMCoeffDecoder', 'Loss'))

  def _nodes(self, which, keys):
    g = _Graph(self, which)
    return {k: Node(g, k) for k in keys}

  def current_lr(self):
    return self.learning_rate * self.decay_rate ** (self.global_step // self.decay_steps)   # staircase (bfmnet.py:308)

  def load_params(self, params):
    self.engine.load_params(params)

  def init_variables(self, seed=None, target=None):
    """Random stand-in weights (xavier-like kernels, unit moving variance) when no checkpoint is available."""
    (target or self.engine).load_params(random_variables(seed))

  # ---- checkpoints: everything tf.train.Saver(var_list=tf.global_variables()) holds (train_bfmnet.py:141-145) ------------------------
  BETA1, BETA2 = 0.9, 0.999

  def _state_dict(self):
    eng = self.train_engine
    d = eng.get_params()
    for tag, buf in (('Adam', eng.m), ('Adam_1', eng.v)):
      host = buf.detach().cpu().numpy()
      off = 0
      for n, shape in eng.trainables():
        k = int(np.prod(shape))
        d['%s/%s' % (n, tag)] = host[off:off + k].reshape(shape).copy()
        off += k
    d['beta1_power'] = np.float32(self.BETA1 ** (eng.step_t + 1))
    d['beta2_power'] = np.float32(self.BETA2 ** (eng.step_t + 1))
    d['global_step'] = np.int32(self.global_step)
    return d

  def save(self, path):
    """.npz -> one numpy archive; anything else -> a TensorFlow V2 checkpoint prefix ('ckpt_bfmnet/bfmnet-65000')."""
    d = self._state_dict()
    if path.endswith('.npz'):
      d['adam_t'] = np.int64(self.train_engine.step_t)
      np.savez(path, **d)
      return path
    from ..utils import tf_checkpoint
    return tf_checkpoint.write_checkpoint(path, d)

  def restore(self, path):
    """tf.train.Saver().restore(sess, 'ckpt_bfmnet/bfmnet-65000') (infer_bfmnet.py:192, infer_bfmvid.py:217, train_bfmnet.py:96-98): a
    TensorFlow checkpoint prefix / directory, or an .npz keyed by the same variable names.  With a training graph built the Adam
    slots, beta powers and global_step are restored too when the checkpoint holds them."""
    from ..audio import bfmnet_manifest
    names = [n for n, _, _ in bfmnet_manifest()]
    if path.endswith('.npz'):
      z = np.load(path)
      d = {k: z[k] for k in z.files}
    else:
      from ..utils import tf_checkpoint
      d = tf_checkpoint.read_checkpoint(path)
    missing = [n for n in names if n not in d]
    if missing:
      raise KeyError('%s holds no %s (%d BFMNet variables missing)' % (path, missing[0], len(missing)))
    if self.engine is not None:
      self.engine.load_params(d)
    eng = self.train_engine
    if eng is not None:
      eng.load_params(d)
      tr = eng.trainables()
      if all(('%s/Adam' % n) in d and ('%s/Adam_1' % n) in d for n, _ in tr):
        import torch
        for tag, buf in (('Adam', eng.m), ('Adam_1', eng.v)):
          flat = np.concatenate([np.asarray(d['%s/%s' % (n, tag)], dtype=np.float32).reshape(-1) for n, _ in tr])
          buf.copy_(torch.from_numpy(flat).to(buf.device))
        if 'adam_t' in d:
          eng.step_t = int(d['adam_t'])
        elif 'beta1_power' in d or 'beta2_power' in d:
          # beta2_power first: float32 0.9 ** (t + 1) underflows to exactly 0 after ~1000 updates (the first save is at 5000),
          # 0.999 ** (t + 1) stays representable for ~100k; both 0 -> 'very many' (the bias correction is 1 by then)
          from ..utils import tf_checkpoint
          eng.step_t = tf_checkpoint.adam_steps_from_beta_powers(d.get('beta1_power', 0), d.get('beta2_power', 0), self.BETA1, self.BETA2)
      else:
        logger.warning('%s holds no Adam slots: the optimiser state starts from zero', path)
      if 'global_step' in d:
        self.global_step = int(d['global_step'])

  def _resolve(self, feeds, feed_dict):
    import torch
    vals, pulled = {}, {}
    for key, node in feeds.items():
      if isinstance(node, IteratorNext):
        if id(node.iterator) not in pulled:
          pulled[id(node.iterator)] = node.iterator.next_batch()
        v = pulled[id(node.iterator)][node.index]
      elif isinstance(node, Placeholder):
        if node not in feed_dict:
          raise ValueError('placeholder %s (%s) was not fed' % (key, node.name))
        v = feed_dict[node]
      elif isinstance(node, Constant):
        v = node.value
      else:
        v = node
      vals[key] = v
    return vals

  def execute(self, names, feed_dict):
    return self._run('infer', names, feed_dict)

  def _run(self, which, names, feed_dict):
    import torch
    feeds = {'infer': getattr(self, '_feeds', None), 'train': getattr(self, '_train_feeds', None), 'eval': getattr(self, '_eval_feeds', None)}[which]
    vals = self._resolve(feeds, feed_dict)
    t = lambda v: (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v, dtype=np.float32))).to('cuda', torch.float32)
    seq = vals['Seq_len']
    seq = (seq.cpu().numpy() if torch.is_tensor(seq) else np.asarray(seq)).astype(np.int32)
    out, aux = {}, {}
    if which == 'train':
      eng = self.train_engine
      lr = self.current_lr()
      apply = 'Train_op' in names
      if any(n in names for n in ('Train_op', 'Loss', 'Grads', 'BFMCoeffDecoder')):
        eng.lr = lr
        if apply:    # small batches: the whole step replays from a hipGraph (dropout draws included); large ones: eager on two streams
          aux = eng.train_step_auto(t(vals['Ears']), t(vals['Mfccs']), t(vals['BFM_coeff_seq']), seq, self.drop_rate)
          self.global_step += 1
        else:
          aux = eng.train_step(t(vals['Ears']), t(vals['Mfccs']), t(vals['BFM_coeff_seq']), seq, masks=eng.draw_masks(self.drop_rate), apply=False)
      coeff = eng.last_out
      aux['Lr'] = np.float32(lr)
    else:
      eng = self.engine if which == 'infer' else self._eval_engine
      if which == 'eval':
        eng.load_params(self.train_engine.get_params())     # the eval graph shares the variables being trained (AUTO_REUSE)
      if which == 'infer' and self.reference_decoder_dropout:
        eng.draw_decoder_dropout(0.25)                      # a fresh draw per run, as a TF session would make
      coeff = eng.forward(t(vals['Ears']), t(vals['Mfccs']), seq)
      if 'Loss' in names:
        aux['loss'] = self.train_engine.eval_loss(coeff, t(vals['BFM_coeff_seq']), seq)
    for n in names:
      if n == 'BFMCoeffDecoder':
        out[n] = coeff.cpu().numpy()
      elif n in ('MfccEncoder', 'RNNModule'):
        out[n] = eng.tensor(n).cpu().numpy()
      elif n == 'Loss':
        out[n] = np.float32(aux['loss'])
      elif n == 'Lr':
        out[n] = aux['Lr']
      elif n == 'Global_step':
        out[n] = self.global_step
      elif n == 'Train_op':
        out[n] = None
      elif n == 'Grads':
        g = self.train_engine.get_grads(unclipped=True)               # as the reference: compute_gradients' outputs, before the clip
        out[n] = [g[k] for k, _ in self.train_engine.trainables()]
      elif n == 'Tvars':
        w = self.train_engine.get_params()
        out[n] = [w[k] for k, _ in self.train_engine.trainables()]
      elif n == 'Seq_len':
        out[n] = seq
      else:
        out[n] = t(vals[n]).cpu().numpy()
    return out


class _Graph(object):
  """One of the graphs built on a BFMNet ('train' / 'eval'); runtime.Session.run groups fetches by owner."""

  def __init__(self, net, which):
    self.net, self.which = net, which

  def execute(self, names, feed_dict):
    return self.net._run(self.which, names, feed_dict)
