#!/bin/bash
for f in "" -DSV_ABL_NODIAG -DSV_ABL_NOPANEL -DSV_ABL_NOTRAIL -DSV_ABL_NOSUBST "-DSV_ABL_NODIAG -DSV_ABL_NOPANEL -DSV_ABL_NOTRAIL -DSV_ABL_NOSUBST"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" 2>&1 | grep -i " error"
  cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python $GRAFT_REPO_ROOT/bench.py --arch iv --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/prof_$$/*kernel_stats.csv")[0])):
    if 'solve' in r['Name']: print("$f  %-20s avg %.1f us"%(r['Name'][:20], float(r['AverageNs'])/1e3))
PY
  rm -rf /tmp/prof_$$; cd $GRAFT_REPO_ROOT
done
