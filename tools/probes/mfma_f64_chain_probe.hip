// v_mfma_f64_16x16x4_f64: issue interval against the number of INDEPENDENT accumulators a wave alternates between,
// with 1 and 2 waves per SIMD (design data for the accumulation loop of k_iv_solve_ll: how many tiles must a wave
// keep in flight before the f64 matrix pipe is busy).
// hipcc --offload-arch=gfx950 -O3 -o mfma_f64_chain_probe mfma_f64_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(512) void probe(double *out, int iters) {
  const int lane = threadIdx.x & 63;
  d4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1e-3 * lane, b = 1.0 + 1e-6 * lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(int threads) {
  static double *out = nullptr;
  if (!out) (void)hipMalloc(&out, sizeof(double) * 256 * 512);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  probe<CH><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<CH><<<256, threads>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16 * CH;  // MFMAs per wave
  const double waves_per_simd = threads / 256.0;
  printf("%d accumulators, %.0f wave(s)/SIMD: %7.1f ns per MFMA of a wave, %6.1f ns per MFMA of the SIMD, %5.1f TFLOP/s\n", CH,
         waves_per_simd, ms * 1e6 / n, ms * 1e6 / (n * waves_per_simd), 2048.0 * n * (threads / 64) * 256 / (ms * 1e-3) / 1e12);
}
int main() {
  run<1>(256); run<1>(256); run<2>(256); run<4>(256); run<8>(256);
  run<1>(512); run<2>(512); run<4>(512); run<8>(512);
  return 0;
}
