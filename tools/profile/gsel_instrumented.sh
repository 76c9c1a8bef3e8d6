#!/bin/bash
# Instrumented build of k_gsel_w (gmm_wide_kernel.hip, -DFB_FXW_STAMP) next to the product library: wall_clock64 stamps of
# one workgroup's phases.  usage (on the GPU box, after `python -m fakebob_amd.build`): tools/profile/gsel_instrumented.sh [outdir]
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/gselw}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
OBJS=$(ls fakebob_amd/build/*.o | grep -v gmm_wide_kernel)
/opt/rocm/bin/hipcc $FLAGS -DFB_FXW_STAMP -c fakebob_amd/csrc/gmm_wide_kernel.hip -o "$OUT/wide_STAMP.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_STAMP.so" "$OUT/wide_STAMP.o" $OBJS
FAKEBOB_HIP_LIB="$(realpath $OUT)/lib_STAMP.so" python tools/profile/gsel_instrumented.py | tee "$OUT/gselw_STAMP.txt"
