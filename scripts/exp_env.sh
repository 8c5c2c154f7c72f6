#!/bin/bash
# A/B of environment knobs on the full step (batch 32 and 4)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/envsweep; mkdir -p $o
run() {
  for gb in 32 4; do
    env "$@" timeout 300 python bench.py --no-cpu-baseline --no-f32 --no-input-pipeline --no-bfmnet-train --no-profile --global-batch $gb > $o/b.json 2> $o/b.err
    python -c "
import json;d=json.load(open('$o/b.json'));print('$* bs$gb',d['ms_per_step'])"
  done
}
run A=0
run VP_WS_CFG=0
run VP_WS_CFG=195
run VP_WS_CFG=3
run VP_PATCH2_MINBLK=128
run VP_PATCH2_MINBLK=1024
run VP_BNSMALL_PG=4096
run VP_BNSMALL_PG=1024
run VP_NO_WSPLIT=1
run A=0
