#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/cfg
timeout 900 python scripts/bench_configs.py > gpurun_out/cfg/configs.jsonl 2> gpurun_out/cfg/err.txt
cat gpurun_out/cfg/configs.jsonl
