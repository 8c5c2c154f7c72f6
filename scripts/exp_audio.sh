#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/audio
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_audio.py tests/test_gpu_api.py -x -q -m gpu -s > $o/pytest.log 2>&1; grep -E "passed|failed|bf16 trunk|Error" $o/pytest.log | tail -6
python3 scripts/bench_audio.py 20 f32
python3 scripts/bench_audio.py 20 bf16
