#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ao; mkdir -p $o
for b in 32 8; do
for nb in 256 384 512 768 1024 1536; do
  echo "== batch $b thin blocks $nb" | tee -a $o/thin2.txt
  python scripts/layer_profile.py $b 256 bf16 tune:thin_blocks_cout8=$nb tune:thin_blocks_dcout8=$nb tune:thin_blocks_cout4=$nb 2>/dev/null | grep -E "cout8|cout4|dcout8" | tee -a $o/thin2.txt
done
done
