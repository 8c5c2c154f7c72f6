#!/bin/bash
# round 5: where dwproj_kernel's time goes - build-time ablations (DWPROJ_ABL: 1 no stencil, 2 no MFMAs, 4 no DMA after the first chunk)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/dwproj_abl; mkdir -p $o
for v in "" dpabl1 dpabl2 dpabl3 dpabl4; do
  lib=""; [ -n "$v" ] && lib="VP_LIB=$GRAFT_REPO_ROOT/voicepuppet_amd/libvp_$v.so"
  rm -rf $o/p
  env $lib timeout 300 rocprofv3 --kernel-trace --stats -d $o/p -o a --output-format csv -- python3 scripts/bench_audio.py 10 > $o/log_$v.txt 2>&1
  f=$(find $o/p -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-full}" | tee -a $o/abl.txt
  [ -n "$f" ] && python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if 'dwproj' in r[0]: print('%-50s calls %s avg %.1f us' % (r[0][9:45], r[1], float(r[3])/1000))
" | sort | tee -a $o/abl.txt
done
rm -rf $o/p
