#!/bin/bash
# round 5: the -m gpu suite, the instrumented count build of k_gmm_fx2w (skip-rate measurement), the driver's bench line
R=$GRAFT_REPO_ROOT; tag=${1:-r05_check}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -v "^\." $O/pytest.log | grep -i "enrolment\|max |err|\|  auto\|  p1\|  p3\|attack of the tail\|passed\|failed\|error\|rc " | tail -40
timeout 600 bash tools/profile/fxw_instrumented.sh gpurun_out/$tag/fxw > $O/fxw.log 2>&1; tail -4 $O/fxw.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_driver_args.json"))
print("value", d["value"], "windows", [round(x, 2) for x in d["config"]["windows_ms"]])
e = d["secondary"].get("end_to_end")
print(json.dumps({k: e[k] for k in ("iterations_per_attack", "calibrated_stop_iterations", "successes", "dynamic_over_static")}))
print("static", e["static"]["wall_s"], e["static"]["stream_busy_s"], "dynamic", e["dynamic"]["wall_s"], e["dynamic"]["stream_busy_s"])
print({k: round(v["value"]) for k, v in d["secondary"].items() if isinstance(v, dict) and "value" in v})
print(d["secondary"].get("error"))
PY
