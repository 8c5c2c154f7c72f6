#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06ag; mkdir -p $o
scripts/probes/hbm_probe > $o/hbm_probe.txt 2>&1
bash scripts/ab.sh -b "32 8 4" "" "VP_LIB=$PWD/voicepuppet_amd/libvp_plainst.so" 2>&1 | grep "^batch" | tee $o/ab.txt
VP_LIB=$PWD/voicepuppet_amd/libvp_plainst.so python scripts/layer_profile.py 32 256 bf16 2>/dev/null > $o/layers_plain.txt
python scripts/layer_profile.py 32 256 bf16 2>/dev/null > $o/layers_nt.txt
python - <<'P'
a={l.split()[0]:float(l.split()[-6]) for l in open('gpurun_out/r06ag/layers_plain.txt') if ' ms ' in l and ':' in l.split()[0]}
b={l.split()[0]:float(l.split()[-6]) for l in open('gpurun_out/r06ag/layers_nt.txt') if ' ms ' in l and ':' in l.split()[0]}
tot=0
for k in sorted(a, key=lambda k:(a[k]-b.get(k,a[k]))):
  d=a[k]-b.get(k,a[k])
  if abs(d)>0.003: print("%-50s plain %.3f nt %.3f  diff %+.3f"%(k,a[k],b.get(k,0),d)); tot+=d
print("sum of diffs", tot)
P
