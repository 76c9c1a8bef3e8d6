#!/bin/bash
# usage: pmc.sh <outname> <counters...>   (separate pass per counter group; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
out=$1; shift
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, collections, glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$out/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:24]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    cnt[(k,r['Counter_Name'])]+=1
for k in agg:
    print(k, {c: round(v/cnt[(k,c)],1) for c,v in agg[k].items()})
PY
