#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06av; mkdir -p $o
for cfg in 6 0 1; do
  echo "== short-K (<=128) launches on cfg $cfg" | tee -a $o/shortk.txt
  python scripts/layer_profile.py 32 256 bf16 tune:igemm_short_k=128 tune:igemm_short_k_cfg=$cfg 2>/dev/null | grep -E "decoder_1:bwd|layer_5:bwd|conv total" | tee -a $o/shortk.txt
done
bash scripts/ab.sh -b "32" "" "tune:igemm_short_k=128 tune:igemm_short_k_cfg=0" "tune:igemm_short_k=128 tune:igemm_short_k_cfg=1" 2>&1 | tee $o/ab.txt
