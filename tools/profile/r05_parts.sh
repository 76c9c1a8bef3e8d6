#!/bin/bash
# which fusions pay with three attacks in flight: FB_FUSE_PARTS (bit 0 front-end, bit 1 finalise + loss, bit 2 update) under bench.py's K = 3
R=$GRAFT_REPO_ROOT; tag=${1:-r05_parts}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for rep in 1 2; do
for m in none 0 1 2 3 4 5 6 7; do
  if [ $m = none ]; then unset FB_FUSE_PARTS; else export FB_FUSE_PARTS=$m; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-single --steps 200 --warmup 20 > $O/b_$m.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$m.json')); print('parts=$m', round(d['value']), d['config'].get('chain'))"
done; done
