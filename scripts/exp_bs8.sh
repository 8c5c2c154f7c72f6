#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/bs8; mkdir -p $o
for t in "overlap=1" "overlap=0" "d_beside_vgg=0" "d_backward_fork=0" "patch3=0"; do
  timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --global-batch 8 --tune $t > $o/b.json 2> $o/b.err
  python -c "
import json;d=json.load(open('$o/b.json'));print('bs8 $t',d['ms_per_step'])"
done
VP_NO_WSPLIT=1 timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --global-batch 8 > $o/b.json 2> $o/b.err
python -c "
import json;d=json.load(open('$o/b.json'));print('bs8 nowsplit',d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --global-batch 8 > $o/b8.json 2> $o/b.err
python - <<'P'
import json
d=json.load(open('gpurun_out/bs8/b8.json'))
print(d['ms_per_step'])
for k in d['kernels'][:14]: print("%-28s calls %5.1f ms %6.3f TF %7.1f"%(k['name'],k['calls_per_step'],k['ms_per_step'],k['tflops']))
P
