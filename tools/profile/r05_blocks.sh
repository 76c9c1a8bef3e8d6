#!/bin/bash
# k_gmm_fx2w's workgroup count (FB_GMM_TARGET_BLOCKS: 256 = one per CU, each ~60 us) with three attacks in flight and alone
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tb in 256 384 512 768; do
  export FB_GMM_TARGET_BLOCKS=$tb
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('target_blocks=$tb', round(d['value']), round(d['single_attack']['value']), d['roofline'].get('solo_launch_ms'), d['roofline'].get('avg_launch_ms'))"
done; done
