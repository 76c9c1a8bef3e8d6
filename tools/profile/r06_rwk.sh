#!/bin/bash
# round 6: which posterior-solve kernel with several attacks in flight -- k_iv_solve_ll (one workgroup per matrix) or
# k_iv_solve_rw with G = 2 / 3 / 5 workgroups per matrix (FB_IV_SOLVE, FB_IV_RW_G); i-vector SV, spd = 50
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_rwk; mkdir -p $O
for k in 2 3 4; do for m in ll rw5 rw3 rw2; do
  unset FB_IV_RW_G; s=ll; [ $m != ll ] && s=rw && export FB_IV_RW_G=${m#rw}
  FB_IV_SOLVE=$s timeout 300 python bench.py --arch iv --steps 40 --warmup 5 --streams $k --no-cpu-baseline --no-single > $O/sv_${m}_$k.json 2>$O/sv_${m}_$k.err
  python -c "
import json;d=json.load(open('$O/sv_${m}_$k.json'));print('streams $k $m: %.0f it/s  windows %s' % (d['value'], d['config']['windows_ms_str']))"
done; done
