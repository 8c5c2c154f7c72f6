"""PixReferNet with the reference's class surface (voicepuppet/pixrefer/pixrefer.py:15-438) on MI355X.

`build_train_op` / `build_inference_op` return the same `nodes` dictionary keys as the reference; the
values are handles executed by voicepuppet_amd.runtime.Session.  The graph itself (generator, three
discriminator applications, VGG-16 trunk, losses, two TF-style Adam optimisers) is the HIP step executor
behind the C ABI (include/vp_hip.h); nothing here computes on the host.
"""
import logging
import os

import numpy as np

from ..builder import ModelBuilder
from ..config.configure import YParams
from ..runtime import Constant, IteratorNext, Node, Placeholder
from ..utils import tf_checkpoint

logger = logging.getLogger(__name__)

TRAIN_KEYS = ['Inputs', 'FGInputs', 'Targets', 'Masks', 'Outputs', 'Alphas', 'Outputs_FG', 'Predict_real', 'Predict_fake',
              'Perceptual_loss', 'Discrim_loss', 'Gen_loss_GAN', 'Gen_loss_L1', 'Gen_loss', 'Global_step', 'Lr', 'Train_op',
              'Discrim_grads_and_vars', 'Gen_grads_and_vars']
INFER_KEYS = ['Inputs', 'FGInputs', 'Targets', 'Outputs', 'Alphas', 'Outputs_FG']
EXTRA_INFER_KEYS = ['Outputs_u8']          # not nodes of the reference: see execute()


class PixReferNet(ModelBuilder):

  def __init__(self, config_path):
    if (not os.path.exists(config_path)):
      logger.error('config_path not exists.')
      exit(0)
    self.__params = PixReferNet.default_hparams(config_path)
    self.engine = None
    self.global_step = 0

  @staticmethod
  def default_hparams(config_path, name='default'):
    params = YParams(config_path, name)
    params.add_hparam('separable_conv', False)
    params.add_hparam('ngf', 64)
    params.add_hparam('ndf', 64)
    params.add_hparam('l1_weight', 500.0)
    params.add_hparam('gan_weight', 1.0)
    params.training['learning_rate'] = 0.0003
    params.training['beta1'] = 0.5
    params.training['decay_rate'] = 0.999
    return params

  @property
  def params(self):
    return self.__params

  def set_params(self, params):
    self.learning_rate = params.training['learning_rate']
    self.beta1 = params.training['beta1']
    self.decay_rate = params.training['decay_rate']
    self.decay_steps = params.training['decay_steps']
    self.batch_size = params.batch_size
    self.separable_conv = params.separable_conv
    if self.separable_conv:
      raise NotImplementedError('separable_conv=True is a dead branch of the reference (pixrefer.py:69-71) and is not built')
    self.ngf = params.ngf
    self.ndf = params.ndf
    self.l1_weight = params.l1_weight
    self.gan_weight = params.gan_weight
    self.dtype = (params.get('amd') or {}).get('dtype', 'bf16')
    self.is_training = params.is_training
    if (params.is_training):
      self.sess = params.sess
      self.vgg_model_path = params.vgg_model_path

  # ---- graph construction ---------------------------------------------------------------------
  def _make_engine(self, height, training):
    from ..engine import PixReferEngine
    self.engine = PixReferEngine(self.batch_size, height, self.ngf, self.ndf, dtype=self.dtype, training=training,
                                 l1_weight=self.l1_weight, gan_weight=self.gan_weight, per_sample_bn=not training)
    self.init_variables()

  def init_variables(self, seed=None):
    """tf.variables_initializer of the non-VGG variables (train_pixrefer.py:89-92): kernels N(0,0.02),
    gamma N(1,0.02), bias/beta 0 (pixrefer.py:64,68,100-101); VGG restored from vgg_model_path when it is
    the TensorFlow checkpoint itself (utils/tf_checkpoint.py) or an .npz of TF-named arrays, else He-normal."""
    rng = np.random.default_rng(seed)
    p = {}
    for which in (0, 1):
      if self.engine.arena(which) is None:
        continue
      for name, _, shape in self.engine.manifests[which]:
        if name.endswith('kernel'):
          p[name] = rng.normal(0, 0.02, shape).astype(np.float32)
        elif name.endswith('gamma'):
          p[name] = rng.normal(1.0, 0.02, shape).astype(np.float32)
        else:
          p[name] = np.zeros(shape, np.float32)
    if self.engine.arena(2) is not None:
      path = getattr(self, 'vgg_model_path', None)
      npz = None
      if tf_checkpoint.is_tf_checkpoint(path):   # the slim vgg_16.ckpt download itself (pixrefer.py:325-327), V1 file or V2 bundle
        npz = tf_checkpoint.read_checkpoint(path, names=[name for name, _, _ in self.engine.manifests[2]])
      for cand in (path, (path or '') + '.npz'):
        if npz is None and cand and os.path.exists(cand) and cand.endswith('.npz'):
          npz = np.load(cand)
      for name, _, shape in self.engine.manifests[2]:
        if npz is not None and name in npz:
          p[name] = npz[name].astype(np.float32)
        elif name.endswith('weights'):
          p[name] = rng.normal(0, np.sqrt(2.0 / (shape[0] * shape[1] * shape[2])), shape).astype(np.float32)
        else:
          p[name] = np.zeros(shape, np.float32)
      if npz is None:
        logger.warning('vgg_16 weights not found at %s: perceptual trunk uses He-normal stand-in weights', path)
    self.engine.load_params(p)

  def build_network(self, inputs, fg_inputs, targets, trainable=True):
    raise NotImplementedError('the network is built inside libvp_hip.so; use build_train_op / build_inference_op')

  def _bind(self, keys, feeds, training):
    h = None
    for f in feeds.values():
      if f is not None and f.shape is not None and len(f.shape) == 4:
        h = int(f.shape[1])
        break
    if h is None:
      raise ValueError('cannot infer the image size from the input nodes')
    self._feeds = feeds
    self._make_engine(h, training)
    return {k: Node(self, k) for k in keys}

  def build_train_op(self, inputs, fg_inputs, targets, masks):
    return self._bind(TRAIN_KEYS, {'Inputs': inputs, 'FGInputs': fg_inputs, 'Targets': targets, 'Masks': masks}, True)

  def build_inference_op(self, inputs, fg_inputs, targets):
    return self._bind(INFER_KEYS + EXTRA_INFER_KEYS, {'Inputs': inputs, 'FGInputs': fg_inputs, 'Targets': targets}, False)

  # ---- execution (called by runtime.Session.run) --------------------------------------------------
  def _resolve(self, feed_dict):
    import torch
    vals, pulled = {}, {}
    for key, node in self._feeds.items():
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
      vals[key] = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v, dtype=np.float32))
      vals[key] = vals[key].to(self.engine.device, torch.float32)
    return vals

  def current_lr(self):
    return self.learning_rate * self.decay_rate ** (self.global_step // self.decay_steps)   # staircase

  def execute(self, names, feed_dict):
    eng = self.engine
    v = self._resolve(feed_dict)
    lr = self.current_lr()
    if 'Train_op' in names:
      eng.train_step(v['Inputs'], v['FGInputs'], v['Targets'], v['Masks'], lr, self.beta1, group=getattr(self, 'group', None))
      self.global_step += 2          # both apply_gradients increment global_step (pixrefer.py:400,407)
    else:
      eng.forward(v['Inputs'], v['FGInputs'], v['Targets'], v.get('Masks'))
    out = {}
    LOSS_KEYS = ('Perceptual_loss', 'Discrim_loss', 'Gen_loss_GAN', 'Gen_loss_L1', 'Gen_loss')
    # (reading the losses waits for the step: only when one of them is fetched)
    losses = eng.losses() if (eng.training and any(n in LOSS_KEYS for n in names)) else {}
    for n in names:
      if n in ('Inputs', 'FGInputs', 'Targets', 'Masks'):
        out[n] = v[n].cpu().numpy()
      elif n in ('Outputs', 'Outputs_u8', 'Alphas', 'Outputs_FG'):
        # formed on the device by one kernel of the library (vp_pixrefer_fetch): deprocess (pixrefer.py:424); 'Outputs_u8' - not a node of the
        # reference - is (Outputs * 255).astype(uint8) for callers that only write the frames out (infer_bfmvid.py:243): a quarter of the
        # device-to-host bytes, same float32 arithmetic, same bytes; 'Alphas' tiled to three channels (pixrefer.py:284); 'Outputs_FG' with
        # the quirk of build_inference_op on an inference plan: deprocess(Outputs_FG + Alphas - 1) (pixrefer.py:436)
        out[n] = eng.fetch(n).cpu().numpy()
      elif n in ('Predict_real', 'Predict_fake'):
        p = eng.tensor('Predict')
        out[n] = p[0 if n == 'Predict_real' else 1].unsqueeze(-1).cpu().numpy()
      elif n in losses:
        out[n] = np.float32(losses[n])
      elif n == 'Global_step':
        out[n] = self.global_step
      elif n == 'Lr':
        out[n] = np.float32(lr)
      elif n == 'Train_op':
        out[n] = None
      elif n in ('Discrim_grads_and_vars', 'Gen_grads_and_vars'):
        which = 1 if n.startswith('Discrim') else 0
        g = eng.get_params(which, src=eng.grads_d if which else eng.grads_g)
        w = eng.get_params(which)
        out[n] = [(g[k], w[k]) for k, _, _ in eng.manifests[which]]
      else:
        raise KeyError(n)
    return out

  # ---- checkpoints: TensorFlow Saver V2 bundles under the reference's variable names, or .npz with the same keys ----------
  BETA2 = 0.999

  def _state_dict(self):
    """Everything tf.train.Saver(var_list=tf.global_variables()) holds for this graph (train_pixrefer.py:150): variables,
    Adam slots '<var>/Adam', '<var>/Adam_1', the optimisers' beta powers, the never-updated batch-norm moving statistics,
    the frozen vgg_16 trunk and global_step."""
    eng = self.engine
    d = {}
    for which in (0, 1):
      d.update(eng.get_params(which))
      for tag, idx in (('Adam', 0), ('Adam_1', 1)):
        st = eng.get_params(which, src=eng.adam['g' if which == 0 else 'd'][idx])
        d.update({'%s/%s' % (k, tag): v for k, v in st.items()})
      for name, _, shape in eng.manifests[which]:
        if name.endswith('batch_normalization/gamma'):    # tf.layers.batch_normalization(training=True): moving stats stay at init
          d[name[:-5] + 'moving_mean'] = np.zeros(shape, np.float32)
          d[name[:-5] + 'moving_variance'] = np.ones(shape, np.float32)
    d.update(eng.get_params(2))
    for scope, t in (('discriminator_train', eng.t_d), ('generator_train', eng.t_g)):
      d[scope + '/beta1_power'] = np.float32(self.beta1 ** (t + 1))
      d[scope + '/beta2_power'] = np.float32(self.BETA2 ** (t + 1))
    d['global_step'] = np.int32(self.global_step)
    return d

  def save(self, path):
    """`path` ending in .npz: one numpy archive (+ the exact update counters); anything else: a TensorFlow V2 checkpoint prefix
    ('ckpt_pixrefer/pixrefernet-20000' -> .index / .data-00000-of-00001 / checkpoint), as the reference's Saver writes."""
    d = self._state_dict()
    if path.endswith('.npz'):
      d['adam_t'] = np.int64([self.engine.t_g, self.engine.t_d])
      np.savez(path, **d)
      return path
    return tf_checkpoint.write_checkpoint(path, d)

  def restore(self, path):
    """Everything save() wrote - parameters, both Adam slot sets, update counters, global_step - from an .npz, a TensorFlow
    checkpoint prefix or a checkpoint directory (infer_bfmvid.py:218, train_pixrefer.py:96-99): a resumed run continues bit
    for bit."""
    if path.endswith('.npz'):
      z = np.load(path)
      d = {k: z[k] for k in z.files}
    else:
      d = tf_checkpoint.read_checkpoint(path)
    missing = [n for n, _, _ in self.engine.manifests[0] if n not in d]
    if missing:
      raise KeyError('%s holds no %s (%d generator variables missing)' % (path, missing[0], len(missing)))
    self.engine.load_params(d)
    if self.engine.training:
      t_g = t_d = None
      if 'adam_t' in d:
        t_g, t_d = int(d['adam_t'][0]), int(d['adam_t'][1])
      else:
        def steps(cands):
          for b1, b2 in cands:
            if b2 in d:
              return tf_checkpoint.adam_steps_from_beta_powers(d.get(b1, 0.0), d[b2], self.beta1, self.BETA2)
          return None
        # discriminator optimiser is created first (pixrefer.py:398,405): its non-slot variables carry no numeric suffix
        t_d = steps([('discriminator_train/beta1_power', 'discriminator_train/beta2_power'), ('beta1_power', 'beta2_power')])
        t_g = steps([('generator_train/beta1_power', 'generator_train/beta2_power'), ('beta1_power_1', 'beta2_power_1')])
      self.engine.load_adam(d, t_g=t_g, t_d=t_d)
    if 'global_step' in d:
      self.global_step = int(np.asarray(d['global_step']).reshape(-1)[0])
