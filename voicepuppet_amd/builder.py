class ModelBuilder(object):
  """Interface of the reference's model classes (voicepuppet/builder.py:1-10)."""

  def __init__(self):
    raise NotImplementedError('__init__ not implemented.')

  def build_network(self):
    raise NotImplementedError('build_network not implemented.')

  def __call__(self):
    raise NotImplementedError('__call__ not implemented.')
