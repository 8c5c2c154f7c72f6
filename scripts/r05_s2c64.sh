#!/bin/bash
# round 5: conv_s2c64.hip - parity (op cases, in situ), layer times, step A/B against the knob off and against the round-4 library
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/s2c64; mkdir -p $o
timeout 900 python -m pytest -x -q --timeout 600 tests/test_gpu_ops.py -k "stride2_conv or more_kernel_classes" tests/test_gpu_step.py -k "stride2 or register_resident or overlapped or more_kernel or stride2_conv" > $o/tests.log 2>&1; tail -15 $o/tests.log
timeout 300 python scripts/layer_profile.py 2>/dev/null | grep -E "conv total|layer_2:fwd|encoder_2:fwd|encoder_fg_2:fwd|conv1_1:fwd|layer_1:fwd|encoder_1:fwd|encoder_fg_1:fwd" > $o/layers.txt; cat $o/layers.txt
bash scripts/ab.sh -b "32 8" "" "tune:s2c64=0" "VP_LIB=$GRAFT_REPO_ROOT/voicepuppet_amd/libvp_r4.so" 2>&1 | grep "^batch" | tee $o/ab.txt
