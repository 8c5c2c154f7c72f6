cd $GRAFT_REPO_ROOT
o=gpurun_out/dc64; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py -q -m gpu -k "transposed_conv_classes or step_parity or overlapped or soak" > $o/t1.log 2>&1; tail -5 $o/t1.log
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bwd_data or more_kernel" > $o/t2.log 2>&1; tail -4 $o/t2.log
bash scripts/ab.sh -b "32" "" "tune:dc64=0" 2>&1 | tail -4
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers.txt 2>&1; grep -E "total|dc64|patch2" $o/layers.txt
