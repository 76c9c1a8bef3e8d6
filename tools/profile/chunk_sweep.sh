#!/bin/bash
# attacks in flight x component chunks per strip of k_gmm_fx2w (FB_GMM_TARGET_BLOCKS: 256 -> 4 chunks of 16 tiles on 240
# workgroups, 128 -> 2 x 32 on 120, 64 -> 1 x 64 on 60): fewer chunks repeat less of the per-strip prologue and leave CUs
# to the other attacks' launches.  value of bench.py (headline workload), 8-launch chain.
R=$GRAFT_REPO_ROOT; cd $R
for K in ${KS:-3 4 5 6}; do for tb in 256 128 64; do
  FB_GMM_TARGET_BLOCKS=$tb python bench.py --steps 100 --warmup 10 --streams $K --chain unfused --no-cpu-baseline --no-secondary --no-single 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('K=$K target_blocks=$tb value %.0f ms/step %.4f gmm avg %.1f us solo %.1f us' % (d['value'], d['ms_per_step'], 1e3*r['avg_launch_ms'], 1e3*r['solo_launch_ms']))"
done; done
