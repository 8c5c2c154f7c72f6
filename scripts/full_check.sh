#!/bin/bash
# the driver's round-end sequence: every -m gpu test, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/full_check
mkdir -p $o
timeout 2400 python -m pytest tests/ -x -q -m gpu > $o/pytest.log 2>&1
grep -E "passed|failed|error" $o/pytest.log | tail -3
grep -E "^E  |Error" $o/pytest.log | head -10
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $o/bench.json 2> $o/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/full_check/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','dtype','n_gpus')})
print('roofline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k!='traffic_source'})
print('f32', d['f32']['ms_per_step'], d['f32']['roofline']['kernel'], round(d['f32']['roofline']['frac'],3), 'step_frac', round(d['f32']['roofline'].get('step_frac',0),3))
print('pcie', d['with_input_pipeline']['ms_per_step'], 'bfmnet_train', d['bfmnet_train']['ms_per_step'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for k in d['kernels'][:14]: print("%-32s calls %5.1f ms %6.3f TF %7.1f"%(k['name'],k['calls_per_step'],k['ms_per_step'],k['tflops']))
print('conv launches', d['conv_launches_per_step'])
P
