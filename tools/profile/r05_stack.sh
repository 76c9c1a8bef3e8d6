#!/bin/bash
# three (and four) attacks in flight: which fusions with the split front-end kernel NOT padded to one workgroup per CU (FB_VADP_STACK=1)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "none 0 3" "5 1 3" "7 1 3" "3 1 3" "5 1 4" "none 0 4" "5 1 2" "7 0 2"; do
  set -- $cfg
  if [ $1 = none ]; then unset FB_FUSE_PARTS; else export FB_FUSE_PARTS=$1; fi
  if [ $2 = 1 ]; then export FB_VADP_STACK=1; else unset FB_VADP_STACK; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-single --steps 200 --warmup 20 --streams $3 --chain unfused 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('parts=$1 stack=$2 K=$3', round(d['value']))"
done; done
