#!/bin/bash
# round-5 development loop: the step / soak / full-width parity tests, then the step time against the end-of-round-4 library
# (voicepuppet_amd/libvp_r4.so, built from commit dafe131) on the same box.  usage (via gpurun): bash scripts/r05_check.sh ["batches"]
cd $GRAFT_REPO_ROOT
o=gpurun_out/check; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_fullwidth.py tests/test_gpu_coverage.py -x -q -m gpu > $o/t1.log 2>&1; tail -4 $o/t1.log
bash scripts/ab.sh -b "${1:-32}" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r4.so" 2>&1 | tail -8
