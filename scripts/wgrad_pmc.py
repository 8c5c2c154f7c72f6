"""Isolated weight-gradient launches for PMC diagnosis: python scripts/wgrad_pmc.py [reps].  Cases = the heaviest wgrad layers of the step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from voicepuppet_amd import _lib
import gpu_util as gu

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CASES = [("conv1_2", 64, 256, 64, 64, 3), ("conv2_2", 64, 128, 128, 128, 3), ("conv3_2", 64, 64, 256, 256, 3), ("dec128", 32, 128, 128, 128, 3)]
L = _lib.lib()
for name, n, h, cin, cout, k in CASES:
  d = gu.conv_desc(0, n, h, h, cin, cout, k, 1, 1, "bf16")
  ho, wo = gu.out_hw(d)
  x = (torch.randn(n, h, h, cin, device="cuda") * 0.5).to(torch.bfloat16)
  dy = (torch.randn(n, ho, wo, cout, device="cuda") * 0.5).to(torch.bfloat16)
  dw = torch.empty(k, k, cin, cout, device="cuda")
  ws = gu.workspace(d)
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
  for i in range(reps + 2):
    if i == 2: ev[0].record()
    _lib.check(L.vp_conv_bwd_weight(ctypes.byref(d), gu.ptr(x), gu.ptr(None), gu.ptr(None), gu.ptr(dy), gu.ptr(dw), gu.ptr(ws), gu.stream()))
  ev[1].record(); torch.cuda.synchronize()
  ms = ev[0].elapsed_time(ev[1]) / reps
  fl = 2.0 * n * ho * wo * cout * k * k * cin
  print("%-8s %.3f ms (incl. slab reduce) %.0f TF" % (name, ms, fl / ms / 1e9))
