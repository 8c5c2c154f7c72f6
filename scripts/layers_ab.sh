#!/bin/bash
# per-layer conv times with and without the few-pixel kernel; usage: layers_ab.sh "4 32"
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/layers_ab
mkdir -p $o
for gb in ${1:-4 32}; do
  VP_SMALLP=0 python scripts/layer_profile.py $gb 256 bf16 > $o/off_$gb.txt 2>&1
  python scripts/layer_profile.py $gb 256 bf16 > $o/on_$gb.txt 2>&1
  echo "== batch $gb: off / on"; head -1 $o/off_$gb.txt | tail -1; grep "conv total" $o/off_$gb.txt $o/on_$gb.txt
  grep -E "^merged_(encoder|decoder)_[2-5]:" $o/on_$gb.txt | sort | awk '{print $1, $2, $3}' > $o/on.s
  grep -E "^merged_(encoder|decoder)_[2-5]:" $o/off_$gb.txt | sort | awk '{print $1, $2, $3}' > $o/off.s
  join $o/off.s $o/on.s | column -t
done
