#!/bin/bash
# the same sweep for the i-vector chain (configs[2]) with three attacks in flight
R=$GRAFT_REPO_ROOT; tag=${1:-r05_parts_iv}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for rep in 1 2; do
for m in none 0 1 4 5; do
  if [ $m = none ]; then unset FB_FUSE_PARTS; else export FB_FUSE_PARTS=$m; fi
  timeout 300 python bench.py --arch iv --no-cpu-baseline --no-secondary --no-single > $O/b_$m.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$m.json')); print('iv parts=$m', round(d['value']))"
done; done
