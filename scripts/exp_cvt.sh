#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/cvt; mkdir -p $o
timeout 1500 python -m pytest tests/ -x -q -m gpu > $o/pytest.log 2>&1; grep -E "passed|failed|Error" $o/pytest.log | tail -3
timeout 300 python scripts/layer_profile.py 32 256 bf16 2>/dev/null | grep -E "conv total"
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
