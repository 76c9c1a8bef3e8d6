#!/bin/bash
# round 6: the -m gpu suite, the driver's bench line, the i-vector bench line
R=$GRAFT_REPO_ROOT; tag=${1:-r06_check}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_driver_args.json"))
print("value", d["value"], "windows", d["config"].get("windows_ms"))
print({k: round(v["value"]) for k, v in d["secondary"].items() if isinstance(v, dict) and "value" in v})
print("single", d.get("single_attack"))
print(d["secondary"].get("error"))
print(json.dumps(d["roofline"]))
PY
timeout 600 python bench.py --arch iv --steps 30 --warmup 5 --no-cpu-baseline > $O/iv.json 2>$O/iv.err
python - $O/iv.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("iv value %.0f single %.0f (%.3f ms) solve %.1f us" % (d["value"], d["single_attack"]["value"], d["single_attack"]["ms_per_step"], 1e3*d["roofline_solve"]["avg_launch_ms"]))
PY
