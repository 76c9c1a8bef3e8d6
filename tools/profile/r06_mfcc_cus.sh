#!/bin/bash
# round 6: k_mfcc_f32 held to a part of the chip on a shared GPU (FB_MFCC_CUS: compute units its launch may take; 128 = the default
# with three or more attacks per GPU, 256 = all): 3 attacks in flight, 200-step windows twice, then the driver's arguments
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_mcus; mkdir -p $O
for c in 256 192 128 96 64; do for rep in 1 2; do
  FB_MFCC_CUS=$c python bench.py --steps 200 --warmup 20 --streams 3 --no-cpu-baseline --no-secondary --no-single > $O/b_$c.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/b_$c.json'));print('FB_MFCC_CUS=$c: %.0f it/s' % d['value'])"
done; done
for c in 256 128; do
  FB_MFCC_CUS=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-single > $O/d_$c.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/d_$c.json'));print('driver args, FB_MFCC_CUS=$c: %.0f it/s  %s' % (d['value'], d['config']['windows_ms_str']))"
done
