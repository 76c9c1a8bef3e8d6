#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --streams 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python - <<PY
import csv,glob,json
for r in csv.DictReader(open(glob.glob("/tmp/prof_$$/*kernel_stats.csv")[0])):
    print("  %-30s avg %.1f us"%(r['Name'][:30], float(r['AverageNs'])/1e3))
print("  it/s", json.load(open('/tmp/b.json'))['value'])
PY
cd $GRAFT_REPO_ROOT; timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
