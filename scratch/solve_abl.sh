#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 8 16 31; do
  FB_IV_SOLVE_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps$d -o iv -- python $GRAFT_REPO_ROOT/bench.py --arch iv --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python -c "
import csv
for r in csv.DictReader(open('/tmp/ps$d/iv_kernel_stats.csv')):
    if 'k_iv_solve' in r['Name']: print('dbg=$d', round(float(r['AverageNs'])/1e3,1))"
done
