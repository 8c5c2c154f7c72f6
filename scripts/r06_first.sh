#!/bin/bash
# round 6, first GPU call: the suite on the round's first build, the bench line with its new sub-records, the hipGraph root-cause experiment
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06a; mkdir -p $o
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc $?" >> $o/pytest.log; tail -5 $o/pytest.log
( time python bench.py ) > $o/bench.json 2> $o/bench.err; tail -3 $o/bench.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r06a/bench.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'step_frac', d['roofline']['step_frac'])
for k in ('bs8_256','h512_bs2','h512_bs8','config3'):
  print(k, {a:b for a,b in d[k].items() if a in ('ms_per_step','step_frac','frames_per_s','bfmnet_ms','logmel_ms')}, d[k].get('dominant_class',{}).get('kernel'))
print('ceiling', d.get('strong_scaling_ceiling'))
P
for b in 4 8; do timeout 600 python scripts/exp_graph2.py $b > $o/graph_bs$b.txt 2>&1; cat $o/graph_bs$b.txt | grep -v amdgpu.ids; done
cp -r gpurun_out/graph $o/ 2>/dev/null
bash scripts/ab.sh -b "32 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_r5.so" 2>&1 | grep -v amdgpu.ids | tee $o/ab.txt
