#!/bin/bash
# ablation of k_gmm_fx2 on the GPU box: what bounds the 132 us?
for f in "" -DFB_FX_DEFER=1 -DFB_ABL_NOEPI -DFB_ABL_NOMFMA -DFB_ABL_NOLOAD -DFB_ABL_NOLDSREAD -DFB_ABL_NOBAR "-DFB_ABL_NOEPI -DFB_ABL_NOLOAD -DFB_ABL_NOLDSREAD -DFB_ABL_NOBAR" "-DFB_ABL_NOMFMA -DFB_ABL_NOLOAD -DFB_ABL_NOLDSREAD -DFB_ABL_NOBAR"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" >/dev/null 2>&1
  echo "== $f"; python scratch/gmm_only.py
done
