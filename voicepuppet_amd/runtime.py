"""A minimal stand-in for the TF1 session surface the reference scripts use.

The reference builds a graph once (`nodes = model.build_train_op(...)`) and then drives it with
`sess.run([nodes['Train_op'], nodes['Gen_loss_GAN'], ...], feed_dict)` (train_pixrefer.py:136-143,
infer_bfmvid.py:221,240).  Here `nodes` holds `Node` handles bound to an executor object; `Session.run`
hands the fetch names and the feeds to that executor, which launches the HIP step and returns numpy.
"""
import numpy as np


class Node(object):
  def __init__(self, owner, name, shape=None):
    self.owner = owner
    self.name = name
    self.shape = shape

  def __repr__(self):
    return "<Node %s of %s>" % (self.name, type(self.owner).__name__)


class Placeholder(Node):
  """tf.placeholder(tf.float32, shape=[None, H, W, C])"""

  def __init__(self, shape, name="Placeholder"):
    Node.__init__(self, None, name, tuple(shape))


def placeholder(shape, name="Placeholder"):
  return Placeholder(shape, name)


class Constant(Node):
  def __init__(self, value, name="Const"):
    Node.__init__(self, None, name, tuple(np.shape(value)))
    self.value = value


def convert_to_tensor(value, name="Const"):
  return value if isinstance(value, Node) else Constant(value, name)


class IteratorNext(Node):
  """One component of `iterator.get_next()`; the executor pulls a fresh batch per run."""

  def __init__(self, iterator, index, shape):
    Node.__init__(self, None, "IteratorGetNext:%d" % index, shape)
    self.iterator = iterator
    self.index = index


class Dataset(object):
  """from_generator(...).shuffle(n).repeat().padded_batch(b) of generator.py:1021-1040."""

  def __init__(self, gen_fn, shapes, batch_size, shuffle_bufsize=0, seed=None):
    self.gen_fn, self.shapes, self.batch_size, self.shuffle_bufsize = gen_fn, shapes, batch_size, shuffle_bufsize
    self.rng = np.random.default_rng(seed)

  def make_one_shot_iterator(self):
    return DatasetIterator(self)


class DatasetIterator(object):
  def __init__(self, ds):
    self.ds = ds
    self._it = None
    self._buf = []

  def _samples(self):
    while True:   # repeat()
      n = 0
      for s in self.ds.gen_fn():
        n += 1
        yield s
      if n == 0:
        raise RuntimeError("the dataset generator yielded nothing")

  def next_batch(self):
    if self._it is None:
      self._it = self._samples()
    out = []
    while len(out) < self.ds.batch_size:
      while len(self._buf) < max(1, self.ds.shuffle_bufsize):
        self._buf.append(next(self._it))
      j = int(self.ds.rng.integers(len(self._buf))) if self.ds.shuffle_bufsize > 1 else 0
      out.append(self._buf.pop(j))
    return tuple(np.stack([s[k] for s in out]).astype(np.float32) for k in range(len(self.ds.shapes)))

  def get_next(self):
    b = self.ds.batch_size
    return tuple(IteratorNext(self, k, (b,) + tuple(s)) for k, s in enumerate(self.ds.shapes))


class Session(object):
  """`sess.run(fetches, feed_dict)`; fetches: a Node or a (nested) list of Nodes."""

  def __init__(self, config=None):
    self.config = config

  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False

  def run(self, fetches, feed_dict=None):
    single = isinstance(fetches, Node)
    flat = [fetches] if single else list(fetches)
    owners = []
    for f in flat:
      if f.owner is None:
        raise ValueError("cannot fetch %r: it is an input node" % (f,))
      if all(f.owner is not o for o in owners):
        owners.append(f.owner)
    results = {}
    for o in owners:
      names = [f.name for f in flat if f.owner is o]
      vals = o.execute(names, feed_dict or {})
      for n in names:
        results[(id(o), n)] = vals[n]
    out = [results[(id(f.owner), f.name)] for f in flat]
    return out[0] if single else out
