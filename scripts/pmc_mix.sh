#!/bin/bash
# instruction mix (VALU / SALU / MFMA / LDS wave-instructions) of every kernel of the training step, single stream
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/pmc_mix
rm -rf $o; mkdir -p $o
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS -d $o/p -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary --tune streams=1 > $o/p.log 2>&1
# second pass (counters alone, like the first): how busy the matrix pipe is - SQ_VALU_MFMA_BUSY_CYCLES against the kernel's GPU-active cycles (GRBM_GUI_ACTIVE x 1024 SIMDs)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $o/q -o q --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-f32 --no-input-pipeline --no-bfmnet-train --no-scaling-ceiling --no-secondary --tune streams=1 > $o/q.log 2>&1
python3 scripts/pmc_mix.py $o/p $o/q $o/mfma_busy.json > $o/mix.txt; head -60 $o/mix.txt
rm -rf $o/p $o/q
