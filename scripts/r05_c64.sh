cd $GRAFT_REPO_ROOT
o=gpurun_out/c64; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_step.py -q -m gpu -k "conv1_2_register or step_parity or overlapped" > $o/t1.log 2>&1; tail -5 $o/t1.log
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_soak.py -q -m gpu -k "overlapped or soak" 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_fwd or bwd_data" > $o/t2.log 2>&1; tail -4 $o/t2.log
bash scripts/ab.sh -b "32" "" "tune:c64=0" 2>&1 | tail -4
VARS="hip" bash scripts/r05_c64_abl.sh
