#!/bin/bash
for C in 32 2048; do
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python $GRAFT_REPO_ROOT/scratch/mfcc_clock.py $C > /tmp/mc.log 2>&1; tail -3 /tmp/mc.log
python - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/prof_$$/*kernel_stats.csv")[0])):
    if 'mfcc' in r['Name'] or 'gmm_fx2' in r['Name'] or 'delta' in r['Name']: print("C=$C  %-28s avg %.1f us"%(r['Name'][:28], float(r['AverageNs'])/1e3))
PY
rm -rf /tmp/prof_$$; cd $GRAFT_REPO_ROOT
done
