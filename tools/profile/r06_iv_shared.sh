#!/bin/bash
# round 6: the i-vector chain's chip-filling launches on a shared GPU -- k_gsel_w with 2 chunks per strip instead of 4
# (FB_GSEL_TARGET_BLOCKS=128), k_mfcc_f32 on 128 / 256 compute units; SV spd = 50, 3 attacks in flight
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_ivsh; mkdir -p $O
for g in 256 128; do for m in 256 128; do
  FB_GSEL_TARGET_BLOCKS=$g FB_MFCC_CUS=$m python bench.py --arch iv --steps 40 --warmup 5 --streams 3 --no-cpu-baseline --no-single > $O/b_${g}_$m.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/b_${g}_$m.json'));print('gsel target $g, mfcc cus $m: %.0f it/s  %s' % (d['value'], d['config']['windows_ms_str']))"
done; done
