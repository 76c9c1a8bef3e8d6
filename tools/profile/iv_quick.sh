#!/bin/bash
# i-vector SV spd=50: solve kernel time + iteration rate, row-wise (default) against FB_IV_SOLVE_LL=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for m in rw ll; do
  export FB_IV_SOLVE=$m
  timeout 300 python bench.py --arch iv --steps 30 --warmup 5 --no-cpu-baseline > $O/iv_$m.json 2>$O/iv_$m.err
  python - $O/iv_$m.json $m <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "value %.0f single %.0f (%.3f ms) solve %.1f us" % (d["value"], d["single_attack"]["value"], d["single_attack"]["ms_per_step"], 1e3*d["roofline_solve"]["avg_launch_ms"]))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
