#!/bin/bash
# A/B of two builds on the same box: bash scripts/exp_ab_lib.sh <other .so under voicepuppet_amd/>
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for lib in libvp_hip.so "$1"; do
    VP_LIB=$GRAFT_REPO_ROOT/voicepuppet_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'])"
  done
done
