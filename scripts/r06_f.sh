#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06f; mkdir -p $o
bash scripts/ab.sh -b "32 8 4" "" "tune:bwd_sums_in_epilogue=0" "tune:bwd_sums_in_epilogue=2" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | grep "^batch" | tee $o/ab.txt
