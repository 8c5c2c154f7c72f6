#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/full
mkdir -p $o
timeout 3000 python -m pytest tests/ -x -q -m gpu > $o/pytest_gpu.log 2>&1
tail -5 $o/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; tail -2 $o/smoke.log
