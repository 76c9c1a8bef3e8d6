#!/bin/bash
# round 6: the i-vector OSI B = 201 line (configs[4]'s share of one GPU) with the threshold gselect and with the dump
R=$GRAFT_REPO_ROOT; tag=${1:-r06_b201}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for m in new dump; do
  unset FB_IV_GSEL_DUMP; [ $m = dump ] && export FB_IV_GSEL_DUMP=1
  timeout 600 python bench.py --arch iv --task OSI --speakers 10 --spd 200 --steps 10 --warmup 3 --no-cpu-baseline > $O/b201_$m.json 2>$O/b201_$m.err
  python - $O/b201_$m.json $m <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "value %.0f single %.0f (%.3f ms) solve %.1f us" % (d["value"], d["single_attack"]["value"], d["single_attack"]["ms_per_step"], 1e3*d["roofline_solve"]["avg_launch_ms"]))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
