#!/bin/bash
for f in "-DFB_R4_WAVES=4 -DFB_R4_OCC=3" "-DFB_R4_WAVES=8 -DFB_R4_OCC=4" "-DFB_R4_WAVES=4 -DFB_R4_OCC=4" "-DFB_R4_WAVES=8 -DFB_R4_OCC=2"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" >/dev/null 2>&1
  echo "== $f"
  python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfcc or feats or score_parity" 2>&1 | tail -2
  cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --streams 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python - <<PY
import csv,glob,json
for r in csv.DictReader(open(glob.glob("/tmp/prof_$$/*kernel_stats.csv")[0])):
    if 'mfcc' in r['Name'] or 'gmm_bx3' in r['Name']: print("  %-30s avg %.1f us"%(r['Name'][:30], float(r['AverageNs'])/1e3))
print("  it/s", json.load(open('/tmp/b.json'))['value'])
PY
  rm -rf /tmp/prof_$$; cd $GRAFT_REPO_ROOT
done
