#!/bin/bash
# Re-create voicepuppet_amd/bfmnet/gemm_tuning/gfx950_batch4_batch32.csv on an MI355X (TunableOp search over the f32 GEMM shapes of the
# BFMNet training step at batch 4 and 32): bash scripts/tune_bfmnet_gemms.sh; the merged file lands in gpurun_out/tunable/merged.csv
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/tunable; mkdir -p $o
python3 - <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voicepuppet_amd.bfmnet.bfmnet import random_variables
from voicepuppet_amd.bfmnet.train_engine import BFMNetTrainEngine
rng = np.random.default_rng(0)
nver, T = 35709, 24
model = {"exBase": rng.normal(0, 0.05, (3 * nver, 64)).astype(np.float32), "vmask": np.ones(3 * nver, np.float32)}
for B in (4, 32):
  eng = BFMNetTrainEngine(B, T, model, tuned_gemms=False)
  eng.load_params(random_variables(0))
  dev = eng.dev
  eng.tune_gemms(torch.rand(B, T, 1, device=dev), torch.randn(B, 5 * T, 80, device=dev), torch.randn(B, T, 257, device=dev), [T] * B,
                 "gpurun_out/tunable/merged.csv")
print(open("gpurun_out/tunable/merged.csv").read().count("\n"), "lines")
P
