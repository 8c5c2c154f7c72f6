#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/audio
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_audio.py -x -q -m gpu > $o/pytest.log 2>&1; tail -2 $o/pytest.log
python3 scripts/bench_audio.py 20
timeout 900 python -m pytest tests/test_gpu_input_pipeline.py -x -q -m gpu > $o/pytest_ip.log 2>&1; tail -2 $o/pytest_ip.log
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-profile > $o/bench.json 2> $o/bench.err
python -c "
import json;d=json.load(open('$o/bench.json'));print(d['ms_per_step'], d['with_input_pipeline'])"
