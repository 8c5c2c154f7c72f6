#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bi; mkdir -p $o
for r in 85 128 170 256; do
  echo "== cout1_bwd rows per group $r" | tee -a $o/rows.txt
  python scripts/layer_profile.py 32 256 bf16 tune:cout1_bwd=$r 2>/dev/null | grep -E "layer_5:bwd" | tee -a $o/rows.txt
done
