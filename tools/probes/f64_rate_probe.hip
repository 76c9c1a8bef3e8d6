// f64_rate_probe.hip -- issue cost of the float64 vector instructions k_mfcc_r16 is made of, per SIMD, with one and two
// waves resident: cycles (s_memtime at the shader clock is not available: wall_clock64, 100 MHz) per instruction for
// chains of 8 independent accumulators.   hipcc --offload-arch=gfx950 -O3 -o f64_rate_probe f64_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 2000
template <int OP>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *ticks) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double b = 1.0000001, c = 1e-12;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float *)&a[i]) : "v"((float)b), "v"((float)c));
      }
  }
  const unsigned long long t1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int OP>
static void run(const char *name, int threads) {
  double *out; unsigned long long *ticks, h;
  hipMalloc(&out, sizeof(double) * 1024 * 256); hipMalloc(&ticks, 8);
  for (int grid : {1, 256}) {
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, ticks);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, ticks);
    hipDeviceSynchronize();
    hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    const double ns = h * 10.0, per = ns / (REP * 32.0) / (threads / 256.0);
    printf("%-10s %3d threads/WG (%d waves/SIMD), %3d WGs: %.2f ns per instruction per SIMD\n", name, threads, threads / 256, grid, per);
  }
  hipFree(out); hipFree(ticks);
}
int main() {
  for (int th : {256, 512}) { run<0>("v_fma_f64", th); run<1>("v_add_f64", th); run<2>("v_mul_f64", th); run<3>("v_fma_f32", th); }
  return 0;
}
