#!/bin/bash
# The ONE parameterised A/B script (replaces the round-1/2 exp_*.sh one-offs; their findings are in DESIGN.md section 3 / 11).
# Step time of the G+D benchmark step under different settings, two rounds each, on the GPU box:
#   bash scripts/ab.sh [-b "4 8 32"] SETTING [SETTING ...]
# a SETTING is a quoted list of environment assignments and / or vp_tune knobs, e.g.
#   bash scripts/ab.sh "" "tune:smallp_max_pixels=0" "tune:patch_tiles=3 tune:streams=1" "VP_LIB=$PWD/voicepuppet_amd/libvp_r4.so"
cd "${GRAFT_REPO_ROOT:-.}"
batches="32"
if [ "$1" = "-b" ]; then batches="$2"; shift 2; fi
o=gpurun_out/ab; mkdir -p $o
for round in 1 2; do
  for gb in $batches; do
    i=0
    for setting in "$@"; do
      i=$((i+1))
      envs=""; tunes=""
      for tok in $setting; do
        case "$tok" in tune:*) tunes="$tunes --tune ${tok#tune:}";; *) envs="$envs $tok";; esac
      done
      env $envs python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-profile --no-f32 --no-scaling-ceiling --no-input-pipeline --no-bfmnet-train --no-secondary --global-batch $gb $tunes > $o/b_${gb}_$i.json 2> $o/b_${gb}_$i.err
      python -c "import json;d=json.load(open('$o/b_${gb}_$i.json'));print('batch $gb [$setting]', round(d['ms_per_step'],3))" || tail -3 $o/b_${gb}_$i.err
    done
  done
done
