#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06n; mkdir -p $o
bash scripts/ab.sh -b "32" "" "tune:vgg_real_fork=1" "tune:vgg_real_fork=2" "tune:vgg_real_fork=4" "tune:d_backward_fork=0" "tune:d_backward_fork=1" "tune:d_beside_vgg=0" 2>&1 | grep "^batch" | tee $o/ab_sched.txt
