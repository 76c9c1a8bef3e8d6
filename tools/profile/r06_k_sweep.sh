#!/bin/bash
# round 6: attacks in flight per GPU with the half-chip launches of a shared GPU (k_gmm_fx2w: FB_GMM_SUB=2, k_mfcc_f32: 128
# compute units -- the chain fb_set_fused_chain(e, 0) runs), 200-step windows
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_ksweep; mkdir -p $O
for k in 2 3 4 5 6; do
  python bench.py --steps 200 --warmup 20 --streams $k --chain unfused --no-cpu-baseline --no-secondary --no-single > $O/b_$k.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/b_$k.json'));print('streams $k: %.0f it/s' % d['value'])"
done
