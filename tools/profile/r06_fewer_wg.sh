#!/bin/bash
# round 6: k_gmm_fx2w with half as many workgroups as compute units, each scoring two component chunks one after the other
# (FB_GMM_SUB=2: the default with three or more attacks per GPU) against one workgroup per compute unit (FB_GMM_SUB=1):
# GMM tests, then 2 - 5 attacks in flight with 200-step windows, then the driver's arguments
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_fewer; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize_gmm.py tests/test_gpu_fenced.py -x -q 2>&1 | tail -2
for sub in 1 2; do for k in 2 3 4 5; do
  FB_GMM_SUB=$sub python bench.py --steps 200 --warmup 20 --streams $k --chain unfused --no-cpu-baseline --no-secondary --no-single > $O/b_${sub}_$k.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/b_${sub}_$k.json'));print('FB_GMM_SUB=$sub, streams $k: %.0f it/s (solo %.1f us)' % (d['value'], 1e3*d['roofline']['solo_launch_ms']))"
done; done
for sub in 1 2; do
  FB_GMM_SUB=$sub python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-single > $O/d_$sub.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/d_$sub.json'));print('driver args, FB_GMM_SUB=$sub: %.0f it/s  %s' % (d['value'], d['config']['windows_ms_str']))"
done
