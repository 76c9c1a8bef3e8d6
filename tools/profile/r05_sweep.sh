#!/bin/bash
# attacks in flight x launch chain, GMM headline workload (100-step windows, median of 5)
R=$GRAFT_REPO_ROOT; tag=${1:-r05_sweep}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for k in 1 2 3 4; do for ch in fused unfused; do
  python bench.py --streams $k --chain $ch --steps 100 --warmup 10 --repeats 5 --no-cpu-baseline --no-secondary --no-single > $O/gmm_k${k}_$ch.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/gmm_k${k}_$ch.json')); print('gmm K=$k $ch', round(d['value']), [round(x,2) for x in d['config']['windows_ms']])"
done; done
for k in 1 2 3 4; do for ch in fused unfused; do
  python bench.py --arch iv --streams $k --chain $ch --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-single > $O/iv_k${k}_$ch.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/iv_k${k}_$ch.json')); print('iv  K=$k $ch', round(d['value']), [round(x,2) for x in d['config']['windows_ms']])"
done; done
