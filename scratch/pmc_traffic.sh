#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --streams 1 --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv, collections, glob, json
res=collections.defaultdict(dict)
for name in ["FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum"]:
    f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_%s/*counter_collection.csv"%name)
    if not f: continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0][:40]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k in agg:
        for c,v in agg[k].items(): res[k][c]=v/cnt[(k,c)]
for k,v in res.items(): print(k, {c: round(x,1) for c,x in v.items()})
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.json","w"), indent=1)
PY
