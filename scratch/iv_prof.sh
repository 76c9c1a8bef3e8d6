#!/bin/bash
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_iv -o p -- python $GRAFT_REPO_ROOT/bench.py --arch iv --steps 20 --warmup 3 --streams 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python - <<PY
import csv,glob,json
tot=0
for r in csv.DictReader(open(glob.glob("/tmp/prof_iv/*kernel_stats.csv")[0])):
    print("  %-34s calls %4s avg %8.1f us  %s%%"%(r['Name'][:34], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
print("  it/s", json.load(open('/tmp/b.json'))['value'])
PY
rm -rf /tmp/prof_iv
