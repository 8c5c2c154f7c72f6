#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ap; mkdir -p $o
for b in 32 4; do
for nb in 256 512 768 1024 2048 4096; do
  echo "== batch $b cin8 blocks $nb" | tee -a $o/cin8.txt
  python scripts/layer_profile.py $b 256 bf16 tune:thin_blocks_cin8=$nb 2>/dev/null | grep -E "cin8" | tee -a $o/cin8.txt
done
done
python scripts/layer_profile.py 32 256 bf16 2>/dev/null | head -40 > $o/layers32.txt
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | tail -12 | tee $o/ab.txt
