// vector f64 fma issue rate on this chip: N independent chains per lane, 2 or 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
  double acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0; for (int i = 0; i < CH; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> void run(int blocks_per_cu) {
  double *d; hipMalloc(&d, sizeof(double) * 256 * 256 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-9, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * CH * iters * 256.0 * grid;
  printf("chains %d, %d waves/SIMD: %.1f TFLOP/s f64 (%.2f ms)\n", CH, blocks_per_cu, flops / ms / 1e9, ms);
  hipFree(d);
}
int main() { run<2>(1); run<4>(1); run<8>(1); run<8>(2); run<8>(4); run<16>(2); return 0; }
