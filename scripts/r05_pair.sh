#!/bin/bash
# round 5: the two-output backward-data of merged2_decoder_2 on conv_s2c64.hip: tests, layer time, step A/B
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/pair; mkdir -p $o
timeout 900 python -m pytest -x -q --timeout 600 tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_coverage.py > $o/tests.log 2>&1; tail -4 $o/tests.log
timeout 300 python scripts/layer_profile.py 2>/dev/null | grep -E "conv total|merged2_decoder_2" > $o/layers.txt; cat $o/layers.txt
bash scripts/ab.sh -b "32 8" "" "tune:s2c64_pair=0" 2>&1 | grep "^batch" | tee $o/ab.txt
