#!/bin/bash
# build variants of k_gmm_fx2 on the GPU box:  FX2_ABL="flagsA|flagsB|..." bash scratch/fx2_abl.sh
IFS='|' read -ra V <<< "${FX2_ABL:-|-DFB_ABL_NOEPI|-DFB_ABL_NOMFMA|-DFB_ABL_NOLOAD|-DFB_ABL_NOLDSREAD|-DFB_ABL_NOBAR}"
for f in "${V[@]}"; do
  FB_EXTRA_HIPCC_FLAGS="$f" python -c "from fakebob_amd import build; build.build(force=True)" >/dev/null 2>&1
  echo "== $f"; python scratch/gmm_only.py
done
