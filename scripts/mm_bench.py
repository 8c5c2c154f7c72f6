"""Per-shape timing of the float32 matrix products of the BFMNet training step (vp_mm_*): python scripts/mm_bench.py [batch] [reps]

Every distinct (P, K, N) of MfccNet's 1x1 convolutions at the given batch (24-frame clips) is run forward, backward-data and
backward-weight; a hipGraph of `reps` launches is replayed so that the host does not limit short kernels.  Prints us and TFLOP/s
per product and the sum weighted by how often the shape occurs in one step.
"""
import ctypes
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from voicepuppet_amd import _lib

_P = ctypes.c_void_p


def ptr(t):
  return _P(t.data_ptr()) if t is not None else _P(0)


def shapes(batch, frames=24):
  """(P, cin, cout, count) of every 1x1 convolution of MfccNet (tinynet.py:159-212) + heads"""
  specs = [(64, 1, False), (64, 6, True), (64, 6, False), (128, 6, True), (128, 6, False), (128, 6, False),
           (192, 6, True), (192, 6, False), (192, 6, False), (192, 6, False), (256, 6, False), (256, 6, False), (256, 6, False),
           (256, 6, True), (256, 6, False), (256, 6, False), (256, 6, False)]
  out = {}
  W, cin = 40, 32
  def add(P, k, n):
    out[(P, k, n)] = out.get((P, k, n), 0) + 1
  for cout, e, pool in specs:
    P = batch * 5 * frames * W
    add(P, cin, cin * e); add(P, cin * e, cout)
    if cout != cin:
      add(P, cin, cout)
    if pool:
      W = (W + 1) // 2
    cin = cout
  add(batch * 5 * frames * W, 256, 256)
  return out


def main():
  batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
  reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
  only = sys.argv[3] if len(sys.argv) > 3 else ""
  L = _lib.lib()
  dev = torch.device("cuda:0")
  st = torch.cuda.Stream()
  tot = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}
  flops_tot = 0.0
  dsz = int(L.vp_mm_pack_desc_bytes())
  with torch.cuda.stream(st):
    sp = _P(st.cuda_stream)
    for (P, K, N), cnt in sorted(shapes(batch).items(), key=lambda kv: (-kv[0][0], kv[0][1], kv[0][2])):
      if only and only != "%dx%dx%d" % (P, K, N):
        continue
      x = torch.randn(P, K, device=dev)
      w = torch.randn(K, N, device=dev) * 0.05
      dy = torch.randn(P, N, device=dev)
      y = torch.empty(P, N, device=dev)
      dx = torch.empty(P, K, device=dev)
      dw = torch.empty(K, N, device=dev)
      ws = torch.zeros(int(L.vp_mm_workspace_bytes(P, K, N)), dtype=torch.uint8, device=dev)
      packed = []
      for d in (0, 1):
        pk = torch.zeros(int(L.vp_mm_packed_bytes(P, K, N, d)), dtype=torch.uint8, device=dev)
        host = ctypes.create_string_buffer(dsz)
        _lib.check(L.vp_mm_pack_desc(0, N, 0, P, K, N, d, 0, host), "desc")
        dd = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
        _lib.check(L.vp_mm_pack_table(ptr(dd), 1, ptr(w), ptr(pk), sp), "pack")
        packed.append(pk)
      ops = {
          "fwd": lambda: L.vp_mm_fwd_f32_packed(ptr(x), K, ptr(packed[0]), _P(0), ptr(y), N, P, K, N, ptr(ws), sp),
          "bwd_data": lambda: L.vp_mm_bwd_data_f32_packed(ptr(dy), N, ptr(packed[1]), ptr(dx), K, 0, P, K, N, ptr(ws), sp),
          "bwd_weight": lambda: L.vp_mm_bwd_weight_f32(ptr(x), K, ptr(dy), N, ptr(dw), P, K, K, N, ptr(ws), sp),
      }
      flops = 2.0 * P * K * N
      line = "%6d x %4d x %4d  (x%d)" % (P, K, N, cnt)
      for name, fn in ops.items():
        _lib.check(fn(), name)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
          for _ in range(reps):
            fn()
        g.replay(); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); g.replay(); g.replay(); e1.record(st); st.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (2 * reps)
        tot[name] += us * cnt
        line += "   %s %7.1f us %5.1f TF" % (name, us, flops / us / 1e6)
      flops_tot += flops * cnt
      # correctness spot check against torch (f32, loose: different summation order)
      ref = x @ w
      err = ((y - ref).abs().max() / ref.abs().max()).item()
      err_dx = ((dx - dy @ w.t()).abs().max() / (dy @ w.t()).abs().max()).item()
      err_dw = ((dw - x.t() @ dy).abs().max() / (x.t() @ dy).abs().max()).item()
      line += "   err %.1e %.1e %.1e" % (err, err_dx, err_dw)
      print(line, flush=True)
  s = sum(tot.values())
  print("batch %d: fwd %.0f us, bwd_data %.0f us, bwd_weight %.0f us, all %.0f us = %.1f TF" %
        (batch, tot["fwd"], tot["bwd_data"], tot["bwd_weight"], s, 3 * flops_tot / s / 1e6))


if __name__ == "__main__":
  main()
