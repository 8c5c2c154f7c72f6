#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06au; mkdir -p $o
python scripts/layer_profile.py 32 256 bf16 2>/dev/null > $o/layers32.txt
python scripts/layer_profile.py 4 256 bf16 2>/dev/null > $o/layers4.txt
wc -l $o/*.txt
