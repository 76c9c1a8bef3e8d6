// two_wave_probe.hip -- does the logsumexp update of k_gmm_fx2w hide behind the MFMAs when a SIMD holds TWO waves with
// one accumulator chain each instead of ONE wave with two chains?  Per gap: one v_mfma_f32_32x32x16_f16 and NV vector
// instructions (half of them v_exp_f32, half v_add_f32), 256 workgroups.  Time per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o two_wave_probe two_wave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP 400
template <int CH, int NV>   // CH = accumulator chains per wave (2: alternate), NV = vector instructions per gap
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *ticks) {
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)0.5f; }
  float e[8], s = 0.f;
  for (int i = 0; i < 8; ++i) e[i] = -1.0f - i;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc[g % CH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g % CH], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV / 2; ++v) {
        asm volatile("v_exp_f32 %0, %1" : "=v"(e[(2 * v) & 7]) : "v"(e[(2 * v + 1) & 7]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(e[(2 * v + 3) & 7]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = wall_clock64();
  float o = s;
  for (int i = 0; i < 16; ++i) o += acc[0][i] + acc[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = o;
  if (threadIdx.x == 0 && blockIdx.x == 8) ticks[0] = t1 - t0;
}
template <int CH, int NV>
static void run(int threads) {
  float *out; unsigned long long *ticks, h;
  hipMalloc(&out, 4 * 512 * 256); hipMalloc(&ticks, 8);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<CH, NV>), dim3(256), dim3(threads), 0, 0, out, ticks);
  hipDeviceSynchronize();
  hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const int waves = threads / 256;
  printf("%d wave(s)/SIMD, %d chain(s)/wave, %2d vector instr per gap: %.1f ns per MFMA per SIMD\n", waves, CH, NV,
         h * 10.0 / (REP * 16.0) / waves);
  hipFree(out); hipFree(ticks);
}
int main() {
  run<2, 0>(256); run<2, 4>(256); run<2, 8>(256); run<2, 12>(256);
  run<1, 0>(512); run<1, 4>(512); run<1, 8>(512); run<1, 12>(512); run<1, 16>(512);
  run<2, 8>(512); run<2, 16>(512);
  return 0;
}
