#!/bin/bash
# round 6: what the arrival ticket of k_gmm_finalize_loss_update (FB_FIN_TICKET=1) costs against roles by blockIdx (default):
# the headline workload with one / two attacks in flight (the fused chain)
cd $GRAFT_REPO_ROOT
for m in blockidx ticket; do
  unset FB_FIN_TICKET; [ $m = ticket ] && export FB_FIN_TICKET=1
  for k in 1 2; do
  python bench.py --steps 100 --warmup 10 --streams $k --no-cpu-baseline --no-secondary --no-single > /tmp/f_$m.json 2>/dev/null
  python -c "
import json;d=json.load(open('/tmp/f_$m.json'));print('$m streams $k: %.0f it/s (%.4f ms/step)' % (d['value'], d['ms_per_step']))"
  done
done
