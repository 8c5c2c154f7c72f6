#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python3 bench.py --steps 30 --warmup 10 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -c 300 gpurun_out/r06_bench.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_bench.json'))
print(d['ms_per_step'], d['strong_scaling_ceiling'])
P
