#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06al; mkdir -p $o
timeout 1800 python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log; grep -E " passed| failed|^FAILED|^ERROR" $o/pytest.log | tail -8
bash scripts/r06_evidence.sh > $o/evidence.log 2>&1; tail -42 $o/evidence.log
