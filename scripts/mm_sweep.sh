#!/bin/bash
# per-product timing of the float32 matrix products under experiment settings: bash scripts/mm_sweep.sh BATCH "ENV..." ["ENV..." ...]
cd "${GRAFT_REPO_ROOT:-.}"
b=${1:-4}; shift
for s in "$@"; do
  echo "== batch $b [$s]"
  env $s python scripts/mm_bench.py $b 10 2>/dev/null | awk '{ if ($1 == "batch") print; else print $1,$3,$5,$7,$8,$12,$13,$17,$18 }'
done
