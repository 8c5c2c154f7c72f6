#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ah; mkdir -p $o
for v in "" c8abl1 c8abl2 c8abl4 c8abl3; do
  if [ -z "$v" ]; then lib=""; else lib="VP_LIB=$PWD/voicepuppet_amd/libvp_$v.so"; fi
  echo "== [$v]"; env $lib python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "cin8|conv total"
done | tee $o/cin8_ablation.txt
