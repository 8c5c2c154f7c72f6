#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ar; mkdir -p $o
bash scripts/ab.sh -b "32 8" "" "tune:vgg_real_reserve_cus=8" "tune:vgg_real_reserve_cus=16" "tune:vgg_real_reserve_cus=32" "tune:vgg_real_reserve_cus=64" "tune:vgg_real_fork=0 tune:vgg_real_reserve_cus=32" 2>&1 | tee $o/ab.txt
