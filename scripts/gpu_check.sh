#!/bin/bash
# round-2 validation pass: GPU suite (step / api / dp) + bench + clean per-layer profile
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/check
mkdir -p $o
timeout 2400 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullwidth.py tests/test_gpu_dp.py -x -q -m gpu > $o/pytest_gpu.log 2>&1
tail -4 $o/pytest_gpu.log
timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers.txt 2>&1; grep "conv total" $o/layers.txt
timeout 600 python bench.py --no-cpu-baseline --no-f32 > $o/bench.json 2> $o/bench.err
VP_NO_OVERLAP=1 timeout 600 python bench.py --no-cpu-baseline --no-f32 --no-profile > $o/bench_nooverlap.json 2>> $o/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-f32 --no-profile --global-batch 4 > $o/bench_bs4.json 2>> $o/bench.err
python - <<'P'
import json
for f in ('bench','bench_nooverlap','bench_bs4'):
  d=json.load(open('gpurun_out/check/%s.json'%f))
  print(f,{k:d[k] for k in ('value','ms_per_step')}, (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))
d=json.load(open('gpurun_out/check/bench.json'))
for k in d['kernels'][:14]: print("%-28s calls %5.1f ms %6.3f TF %7.1f"%(k['name'],k['calls_per_step'],k['ms_per_step'],k['tflops']))
print(sum(k['ms_per_step'] for k in d['kernels']))
P
tail -3 $o/bench.err
