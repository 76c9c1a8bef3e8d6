#!/bin/bash
for f in "" "-DFB_FX_IPB=2" "-DFB_FX_IPB=3" "-DFB_FX_OCC=3" "-DFB_FX_IPB=2 -DFB_FX_OCC=3"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" >/dev/null 2>&1
  echo "== [$f]"; python scratch/gmm_only.py; python scratch/gmm_only.py
  python scratch/bx_err.py 2>&1 | tail -1
done
