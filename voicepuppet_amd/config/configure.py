"""YParams: the hyper-parameter container of the reference (config/configure.py:7-12) without TensorFlow.

The reference subclasses tf.contrib.training.HParams; the call sites only use attribute access,
`add_hparam`, assignment, and dict-valued entries (`params.mel['sample_rate']`, `params.training[...]`).
"""
import yaml


class YParams(object):
  def __init__(self, yaml_fn, config_name):
    self._names = []
    with open(yaml_fn) as fp:
      for k, v in yaml.load(fp, Loader=yaml.FullLoader)[config_name].items():
        self.add_hparam(k, v)

  def add_hparam(self, name, value):
    # HParams.add_hparam raises when the name exists; the reference scripts never re-add a name
    if name in self._names:
      raise ValueError('Hyperparameter name is reserved: %s' % name)
    self._names.append(name)
    object.__setattr__(self, name, value)

  def set_hparam(self, name, value):
    if name not in self._names:
      raise KeyError(name)
    object.__setattr__(self, name, value)

  def __setattr__(self, name, value):
    # the reference also assigns fresh attributes directly (params.batch_size = 2, params.sess = sess)
    if not name.startswith('_') and name not in self._names:
      self._names.append(name)
    object.__setattr__(self, name, value)

  def __contains__(self, name):
    return name in self._names

  def get(self, name, default=None):
    return getattr(self, name, default)

  def values(self):
    return {k: getattr(self, k) for k in self._names}
