#!/bin/bash
# conv_c64.hip: LDS read-ahead depth and build-time ablations, per-layer times of conv1_2 (timing only for the ablation libraries)
cd $GRAFT_REPO_ROOT
o=gpurun_out/c64; mkdir -p $o
for v in ${VARS:-hip xabl3 xabl7 xabl11 xabl15}; do
  echo "== $v"; VP_LIB=$PWD/voicepuppet_amd/libvp_$v.so timeout 300 python scripts/layer_profile.py 32 256 bf16 2>&1 | grep -E "conv1_2|total"
done 2>&1 | tee $o/abl.txt
