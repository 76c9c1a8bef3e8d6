#!/bin/bash
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("/tmp/prof_tl/*kernel_trace.csv")[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find an iteration in the middle
idx=[i for i,r in enumerate(rows) if 'k_perturb' in r['Kernel_Name']]
i0=idx[len(idx)//2]; i1=idx[len(idx)//2+1]
t0=int(rows[i0]['Start_Timestamp'])
prev_end=None
for r in rows[i0:i1+1]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap = (s-prev_end)/1e3 if prev_end else 0
    print("%-28s start %8.1f  dur %7.1f  gap_before %6.1f  grid %s wg %s" % (r['Kernel_Name'][:28], (s-t0)/1e3, (e-s)/1e3, gap, r.get('Grid_Size_X','?'), r.get('Workgroup_Size_X','?')))
    prev_end=e
PY
rm -rf /tmp/prof_tl
