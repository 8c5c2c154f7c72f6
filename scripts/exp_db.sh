#!/bin/bash
# experiment: register-double-buffered big tiles vs the round-1 kernels (per-layer times of a full step)
cd "$GRAFT_REPO_ROOT"
export VP_DB_MINK=256
o=gpurun_out/exp_db
mkdir -p $o
VP_DBTILE=3 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_fwd or conv_bwd_data" > $o/pytest.log 2>&1
tail -5 $o/pytest.log
for cfg in 0 3 1 2; do
  VP_DBTILE=$cfg timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_dbt$cfg.txt 2>&1
  head -1 $o/layers_dbt$cfg.txt
done
VP_DBTILE=3 VP_DB_NST=3 timeout 300 python scripts/layer_profile.py 32 256 bf16 > $o/layers_dbt3_nst3.txt 2>&1
head -1 $o/layers_dbt3_nst3.txt
VP_DBTILE=3 timeout 300 python bench.py --no-cpu-baseline --no-f32 > $o/bench_dbt3.json 2> $o/bench_dbt3.err
python -c "import json;d=json.load(open('$o/bench_dbt3.json'));print(d['ms_per_step'],d['value'],d['roofline'])"
