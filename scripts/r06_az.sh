#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python - <<'P' 2>&1 | grep -v amdgpu.ids | tail -12
import torch
from voicepuppet_amd.engine import PixReferEngine
for n in (8, 2):
  eng = PixReferEngine(n, 256, 64, 64, dtype="bf16", training=True)
  eng.load_params(eng.random_params(6))
  g = torch.Generator(device="cpu").manual_seed(10)
  batch = [torch.rand(n, 256, 256, c, generator=g).cuda() for c in (6, 6, 3, 3)]
  for mode in (1, 2):
    eng.profile(mode)
    eng.forward(*batch); eng.backward()
    torch.cuda.synchronize()
    recs = eng.profile_collect()
    print(n, mode, sorted(r["name"] for r in recs if "layer_5" in r["name"] or "cout1" in r["name"] or "128x128" in r["name"])[:12])
    eng.profile(0)
P
