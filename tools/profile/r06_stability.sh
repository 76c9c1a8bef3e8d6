#!/bin/bash
# round 6: spread of 40-step windows with three attacks in flight -- the half-chip launches (default) against the whole-chip ones
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_stab; mkdir -p $O
for rep in 1 2 3; do for m in half whole; do
  unset FB_GMM_SUB FB_MFCC_CUS; [ $m = whole ] && export FB_GMM_SUB=1 FB_MFCC_CUS=256
  python bench.py --steps 40 --warmup 10 --repeats 15 --precondition 20 --no-cpu-baseline --no-secondary --no-single > $O/b_${m}_$rep.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/b_${m}_$rep.json'));w=d['config']['windows_ms'];print('$m run $rep: median %.0f it/s, windows min %.2f max %.2f ms: %s' % (d['value'], min(w), max(w), ' '.join('%.1f'%x for x in w)))"
done; done
