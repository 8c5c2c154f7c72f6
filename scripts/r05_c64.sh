cd $GRAFT_REPO_ROOT
o=gpurun_out/c64; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py tests/test_gpu_coverage.py -q -m gpu --timeout 300 -k "conv1_2_register or step_parity or overlapped or soak or coverage or transposed" > $o/t1.log 2>&1; tail -5 $o/t1.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu --timeout 300 -k "conv_fwd or bwd_data or tile_variants" > $o/t2.log 2>&1; tail -4 $o/t2.log | cut -c1-200
bash scripts/ab.sh -b "32" "" "tune:c64=0" "VP_LIB=$PWD/voicepuppet_amd/libvp_r4.so" 2>&1 | tail -6
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers.txt 2>&1; grep -E "total|conv1|conv2" $o/layers.txt
