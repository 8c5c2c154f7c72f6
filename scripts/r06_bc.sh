#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do python3 bench.py --steps 30 --warmup 10 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; python -c "import json; d=json.load(open(\"gpurun_out/r06_bench.json\")); print(d[\"ms_per_step\"], d[\"strong_scaling_ceiling\"][\"ms_per_step_of_each_engine\"], d[\"bs8_256\"][\"ms_per_step\"], d[\"h512_bs2\"][\"ms_per_step\"])"; done
tail -c 300 gpurun_out/r06_bench.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_bench.json'))
print(d['ms_per_step'], d['strong_scaling_ceiling'])
P
