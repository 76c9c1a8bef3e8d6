#!/bin/bash
# round 6: attacks in flight x hardware queues (GPU_MAX_HW_QUEUES; HIP's default is 4) with the half-chip launches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_hwq; mkdir -p $O
for q in 4 8; do for k in 3 4 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 200 --warmup 20 --streams $k --chain unfused --no-cpu-baseline --no-secondary --no-single > $O/b_${q}_$k.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/b_${q}_$k.json'));print('GPU_MAX_HW_QUEUES=$q, streams $k: %.0f it/s' % d['value'])"
done; done
