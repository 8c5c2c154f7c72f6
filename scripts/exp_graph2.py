"""Root-causing the hipGraph result of round 5 ("a hipGraph replay of the step is 2.3x slower than eager; no single-stream
capture, per-branch sub-graphs or explicit kernel-node graph was tried").  Timing only (the captured Adam step count is frozen).

  python scripts/exp_graph2.py BATCH [HEIGHT] [MODE ...]

modes (default: all), each printed as one line `mode: eager X ms | replay Y ms | nodes N | per-node replay cost Z us`:
  multi     the shipped four-stream schedule captured into ONE graph (round 5's experiment)
  single    vp_pixrefer_desc::streams = 1: the whole step captured from ONE stream (a linear chain of kernel nodes)
  phases    single-stream capture split into three graphs (forward | backward + update), launched back to back on one stream
  fwdonly   only the forward pass captured (multi-stream) - is it the fork/join structure or the node count that costs?
  trace     (with rocprofv3 --kernel-trace around the call) replay the single-stream graph 20 times so that scripts/timeline.py can
            measure kernel-to-kernel gaps inside a replay
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicepuppet_amd.engine import PixReferEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
h = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 256
modes = [a for a in sys.argv[2:] if not a.isdigit()] or ["multi", "single", "phases", "fwdonly"]
dev = torch.device("cuda", 0)


def timed(fn, steps=40, warm=10):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps * 1e3


def make(streams):
  eng = PixReferEngine(n, h, 64, 64, dtype="bf16", training=True, streams=streams)
  eng.load_params(eng.random_params(seed=0))
  g = torch.Generator(device=dev).manual_seed(0)
  batch = [torch.rand(n, h, h, c, device=dev, generator=g) for c in (6, 6, 3, 3)]
  return eng, batch


def capture(fn, dump=None):
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    for _ in range(3):
      fn()
  torch.cuda.current_stream().wait_stream(side)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  if dump:
    graph.enable_debug_mode()
  with torch.cuda.graph(graph):
    fn()
  if dump:
    try:
      graph.debug_dump(dump)        # hipGraphDebugDotPrint: the captured nodes and dependency edges
    except Exception as e:         # noqa
      print("debug_dump failed:", e)
  return graph


def nodes_of(path):
  """(kernel nodes, edges) of a hipGraphDebugDotPrint file; (-1, -1) if the dump is missing."""
  try:
    txt = open(path).read()
  except OSError:
    return -1, -1
  import re
  return len(re.findall(r"^\s*\"?[\w]+\"?\s*\[", txt, re.M)), txt.count("->")


OUT = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "graph")
os.makedirs(OUT, exist_ok=True)


def report(mode, eager, rep, dot):
  nn, ne = nodes_of(dot)
  per = (rep - eager) / nn * 1e3 if nn > 0 else float("nan")
  print("bs%d %dx%d %-8s eager %.3f ms | replay %.3f ms | nodes %d edges %d | (replay - eager) per node %.2f us" % (n, h, h, mode + ":", eager, rep, nn, ne, per), flush=True)


for mode in modes:
  if mode == "multi":
    eng, b = make(0)
    fn = lambda: eng.train_step(*b, lr=3e-4)
    eager = timed(fn)
    dot = os.path.join(OUT, "multi_bs%d.dot" % n)
    report(mode, eager, timed(capture(fn, dot).replay), dot)
  elif mode == "single":
    eng, b = make(1)
    fn = lambda: eng.train_step(*b, lr=3e-4)
    eager = timed(fn)
    dot = os.path.join(OUT, "single_bs%d.dot" % n)
    report(mode, eager, timed(capture(fn, dot).replay), dot)
  elif mode == "phases":
    eng, b = make(1)
    eng.fused_update = True
    f1 = lambda: eng.forward(*b)
    (m_g, v_g), (m_d, v_d) = eng.adam["g"], eng.adam["d"]
    from voicepuppet_amd import _lib
    from voicepuppet_amd.engine import _ptr, _stream
    f2 = lambda: _lib.check(eng.L.vp_pixrefer_backward_update(eng.h, _ptr(m_g), _ptr(v_g), _ptr(m_d), _ptr(v_d), 1, 1, 3e-4, 0.5, 0.999, 1e-8, _stream()))
    both = lambda: (f1(), f2())
    eager = timed(both)
    g1 = capture(f1)
    g2 = capture(f2)
    report(mode, eager, timed(lambda: (g1.replay(), g2.replay())), os.path.join(OUT, "single_bs%d.dot" % n))
  elif mode == "fwdonly":
    for streams in (0, 1):
      eng, b = make(streams)
      f1 = lambda: eng.forward(*b)
      eager = timed(f1)
      dot = os.path.join(OUT, "fwd_s%d_bs%d.dot" % (streams, n))
      report("fwd/s%d" % streams, eager, timed(capture(f1, dot).replay), dot)
  elif mode in ("trace", "tracemulti"):
    eng, b = make(1 if mode == "trace" else 0)
    fn = lambda: eng.train_step(*b, lr=3e-4)
    timed(fn, 5, 3)
    g = capture(fn)
    timed(g.replay, 20, 5)
  del eng
  torch.cuda.empty_cache()
