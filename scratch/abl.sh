#!/bin/bash
for v in "-DFB_ABL_NOLOAD" "-DFB_ABL_NOLOAD -DFB_ABL_NOSTORE" "-DFB_ABL_NOLOAD -DFB_ABL_NOSTORE -DFB_ABL_NOEPI" "-DFB_ABL_NOLOAD -DFB_ABL_NOSTORE -DFB_ABL_NOEPI -DFB_ABL_NOBAR"; do
  FB_EXTRA_HIPCC_FLAGS="$v" python fakebob_amd/build.py --force >/dev/null 2>&1
  echo -n "variant [$v] "; python scratch/gmm_only.py 2>&1 | tail -1
done
