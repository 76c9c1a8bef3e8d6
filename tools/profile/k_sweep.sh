#!/bin/bash
# attacks in flight x launch chain: value of bench.py (headline workload)
R=$GRAFT_REPO_ROOT; cd $R
for K in 1 2 3 4 5; do for chain in fused unfused; do
  python bench.py --steps 100 --warmup 10 --streams $K --chain $chain --no-cpu-baseline --no-secondary --no-single 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$K $chain value %.0f ms/step %.4f' % (d['value'], d['ms_per_step']))"
done; done
