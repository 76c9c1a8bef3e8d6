#!/bin/bash
for f in "-DFB_BX_OCC=2" "-DFB_BX_OCC=3"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" 2>&1 | grep -i " error"
  echo "== $f"
  python scratch/gmm_only.py
  FB_GMM_TARGET_BLOCKS=768 python scratch/gmm_only.py
  for k in 1 3; do python bench.py --no-cpu-baseline --streams $k | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' streams', d['config']['attacks_in_flight_per_gpu'], round(d['value'],1))"; done
done
