#!/bin/bash
# usage: mfcc_one.sh "<extra hipcc flags>"
FB_EXTRA_HIPCC_FLAGS="$1" python -c "from fakebob_amd import build; build.build(force=True)" 2>&1 | grep -i error
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfcc or feats or score_parity" 2>&1 | tail -2
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --streams 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python - <<PY
import csv,glob,json
for r in csv.DictReader(open(glob.glob("/tmp/prof_$$/*kernel_stats.csv")[0])):
    print("  %-30s avg %.1f us"%(r['Name'][:30], float(r['AverageNs'])/1e3))
print("  it/s", json.load(open('/tmp/b.json'))['value'])
PY
rm -rf /tmp/prof_$$
