#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ak; mkdir -p $o
for v in "" c8n4; do
  if [ -z "$v" ]; then lib=""; else lib="VP_LIB=$PWD/voicepuppet_amd/libvp_$v.so"; fi
  echo "== [$v]"; env $lib python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "cin8|conv total"
done | tee $o/cin8_n4.txt
