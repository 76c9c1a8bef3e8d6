cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_trace; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $R/bench.py --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-secondary --no-single > $O/bench.json 2>/dev/null
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','')) for r in rows]
ev.sort()
# the steady part: the last 40 % of the trace
t0 = ev[int(len(ev)*0.6)][0]; ev = [e for e in ev if e[0] >= t0]
T = ev[-1][1] - ev[0][0]
fx = [e for e in ev if e[2].startswith('k_gmm_fx2w')]
busy = 0; cur_s, cur_e = fx[0][0], fx[0][1]
for s, e, _ in fx[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.1f us, %d fx2w launches: period %.1f us, fx2w running %.1f %% of the time, mean fx2w duration %.1f us" % (T/1e3, len(fx), T/1e3/len(fx), 100.0*busy/T, sum(e-s for s,e,_ in fx)/len(fx)/1e3))
gaps = [(fx[i+1][0] - max(f[1] for f in fx[:i+1][-3:])) / 1e3 for i in range(len(fx)-1)]
import statistics
print("gap between the end of one fx2w and the start of the next: median %.1f us, mean %.1f" % (statistics.median(gaps), sum(gaps)/len(gaps)))
# what runs in a typical gap
import collections
for i in range(5, 9):
    a, b = fx[i][1], fx[i+1][0]
    inside = [(n, (max(s,a)-a)/1e3, (min(e,b)-a)/1e3) for s,e,n in ev if e > a and s < b and not n.startswith('k_gmm_fx2w')]
    print("gap %.1f us:" % ((b-a)/1e3), ", ".join("%s[%.1f-%.1f]" % (n.split('<')[0][2:14], x, y) for n,x,y in inside))
PY
rm -rf $O/t
