#!/bin/bash
# rocprofv3 kernel stats of a 1-attack bench: usage kstats.sh <tag> [bench args...]
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t1 -o p -- python $R/bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline --no-secondary "$@" > $O/b1.json 2>/dev/null
cp $(find $O/t1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/t1
python - $O/kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-60s calls %5s avg %9.1f ns  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
