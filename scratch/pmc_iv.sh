#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$1; shift
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/$out -o p -- python $GRAFT_REPO_ROOT/bench.py --arch iv --steps 5 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, collections, glob
f=glob.glob("/tmp/$out/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:30]
    if not any(t in k for t in ('solve', 'fullcov', 'stats', 'contract', 'select')): continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k in agg: print(k, {c: round(v/cnt[(k,c)]) for c,v in agg[k].items()})
PY
