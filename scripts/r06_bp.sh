#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06bp; mkdir -p $o
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E " passed| failed|Error|FAILED" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-60
for i in 1 2 3; do python3 bench.py > $o/bench$i.json 2> $o/bench$i.err; python -c "import json; d=json.load(open('$o/bench$i.json')); print(round(d['ms_per_step'],3), [round(x,3) for x in d['strong_scaling_ceiling']['ms_per_step_of_each_engine']], round(d['bs8_256']['ms_per_step'],3), round(d['h512_bs2']['ms_per_step'],3), round(d['h512_bs8']['ms_per_step'],3), round(d['with_input_pipeline']['ms_per_step'],3))"; done
