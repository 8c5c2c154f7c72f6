#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ax; mkdir -p $o
timeout 300 scripts/probes/lastblock_probe 2>&1 | tee $o/lastblock_probe.txt
