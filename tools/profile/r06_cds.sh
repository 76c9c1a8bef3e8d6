cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_cds; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
OBJS=$(ls fakebob_amd/build/*.o | grep -v ivector_kernels)
for v in 3 2 4; do
  /opt/rocm/bin/hipcc $FLAGS -DFB_CD_S=$v -c fakebob_amd/csrc/ivector_kernels.hip -o $OUT/k_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$v.so $OUT/k_$v.o $OBJS
  for cfg in "sv:--arch iv --steps 40 --warmup 5 --streams 1" "b201:--arch iv --task OSI --speakers 10 --spd 200 --steps 10 --warmup 3 --streams 1"; do
    n=${cfg%%:*}; a=${cfg#*:}
    FAKEBOB_HIP_LIB=$PWD/$OUT/lib_$v.so python bench.py $a --no-cpu-baseline > $OUT/${n}_$v.json 2>/dev/null
    python -c "
import json;d=json.load(open('$OUT/${n}_$v.json'));c=d['roofline_contraction'];print('FB_CD_S=$v $n: %.0f it/s, contraction %.1f us' % (d['value'], 1e3*c['avg_launch_ms']))"
  done
done
rm -f $OUT/*.o $OUT/*.so
