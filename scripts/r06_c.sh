#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06c; mkdir -p $o
# (1) where the VGG pass of the real half starts (generator layer index; 0 = at once)
bash scripts/ab.sh -b "32 8 4" "" "tune:vgg_real_fork=3" "tune:vgg_real_fork=8" "tune:vgg_real_fork=12" "tune:vgg_real_fork=16" 2>&1 | grep "^batch" | tee $o/ab_vggfork.txt
# (2) the HIP runtime's graph knobs on the multi-stream graph at batch 4 (root cause of replay = 2.3 x eager)
for e in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "AMD_DIRECT_DISPATCH=0"; do
  echo "== env [$e]" | tee -a $o/graph_env.txt
  env $e timeout 300 python scripts/exp_graph2.py 4 multi single 2>&1 | grep "^bs" | tee -a $o/graph_env.txt
done
