#!/bin/bash
# Instrumented builds of k_gmm_fx2w (gmm_wide_kernel.hip) next to the product library, and what they print:
#   count: how often the logsumexp rescue / the reference moves run   (-DFB_FXW_COUNT, fb_debug_fxw_counts)
#   stamp: wall_clock64 stamps of one workgroup's phases               (-DFB_FXW_STAMP, fb_debug_fxw_stamps)
# usage (on the GPU box, after `python -m fakebob_amd.build`):  tools/profile/fxw_instrumented.sh [outdir]
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/fxw}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
OBJS=$(ls fakebob_amd/build/*.o | grep -v gmm_wide_kernel)
for v in COUNT STAMP; do
  /opt/rocm/bin/hipcc $FLAGS -DFB_FXW_$v -c fakebob_amd/csrc/gmm_wide_kernel.hip -o "$OUT/wide_$v.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_$v.so" "$OUT/wide_$v.o" $OBJS
  FAKEBOB_HIP_LIB="$PWD/$OUT/lib_$v.so" python tools/profile/fxw_instrumented.py $v | tee "$OUT/fxw_$v.txt"
done
