#!/bin/bash
# solo k_gmm_fx2w launch time (auto P and forced P=3) + 3-attack throughput
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for p in "" 3; do
  FB_GMM_DELTA_P=$p python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > $O/q_p${p:-auto}.json 2>/dev/null
  python - $O/q_p${p:-auto}.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "tiles", d["config"]["gmm_delta_p"]["tiles_p1"], d["config"]["gmm_delta_p"]["tiles_p3"], "solo_ms %.4f avg_ms %.4f value %.0f single %.0f (%.4f ms)" % (r["solo_launch_ms"], r["avg_launch_ms"], d["value"], d["single_attack"]["value"], d["single_attack"]["ms_per_step"]))
PY
done
