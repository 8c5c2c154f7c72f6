#!/bin/bash
# long lock-step soaks of the final build: four-stream against single-stream schedule, bit for bit after every step
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06_soak; mkdir -p $o
for cfg in "300 32 256" "300 8 256" "300 4 256" "100 2 512"; do
  echo "== soak2.py $cfg" | tee -a $o/soak.txt
  timeout 900 python scripts/soak2.py $cfg 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $o/soak.txt
done
echo "== soak.py 300 8" | tee -a $o/soak.txt
timeout 900 python scripts/soak.py 300 8 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $o/soak.txt
