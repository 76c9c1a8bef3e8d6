#!/bin/bash
# timing of k_mfcc_r4 build variants (valid outputs only: garbage features stop the attack and empty the launches)
IFS='|' read -ra V <<< "${R4_VAR:-}"
for f in "${V[@]}"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" 2>&1 | grep -i " error"
  python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfcc or feats" 2>&1 | tail -1
  cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --streams 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/prof_$$/*kernel_stats.csv")[0])):
    if 'mfcc' in r['Name']: print("[$f]  %-20s avg %.1f us"%(r['Name'][:20], float(r['AverageNs'])/1e3))
PY
  rm -rf /tmp/prof_$$; cd $GRAFT_REPO_ROOT
done
