// micro-probe: the MFMA sequence of k_gmm_fx2w's F6 step on two accumulators -- ten f16 K = 16 MFMAs, then the scaled
// ones -- in several orders: does a change of operand format between dependent MFMAs (f16 -> fp6, fp6 -> fp4 -> fp6) cost
// more than the instructions themselves?  One wave per SIMD, full chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define F16(acc, x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0)
#define FP6(acc, x, y, sa, sb) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, acc, 2, 2, 0, sa, 0, sb)
#define FP4(acc, x, y, sa, sb) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, acc, 4, 2, 0, sa, 0, sb)
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters) {
  extern __shared__ float pad[];
  const int lane = threadIdx.x & 63;
  f32x16 a0, a1, b0, b1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; b0[r] = 0.f; b1[r] = 0.f; }
  f16x8 h1, h2;
#pragma unroll
  for (int i = 0; i < 8; ++i) { h1[i] = (_Float16)(0.001f * (lane + i)); h2[i] = (_Float16)(0.002f * (lane - i)); }
  i32x8 v1, v2, v3;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v1[i] = i < 6 ? 0x08208208 + lane * (i + 1) : 0; v2[i] = i < 6 ? 0x04104104 + 3 * lane * (i + 2) : 0; v3[i] = i < 4 ? 0x11111111 * (i + 1) + lane : 0; }
  const int s1 = 120 + (lane & 7), s2 = 125 - (lane & 3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 5; ++c) { F16(a0, h1, h2); F16(a1, h2, h1); }
    if (MODE == 0) {         // the kernel's order: fp6, fp4, fp6, fp6 on the same accumulators
      FP6(a0, v1, v2, s1, s2); FP6(a1, v1, v2, s1, s2);
      FP4(a0, v3, v2, s1, s2); FP4(a1, v3, v2, s1, s2);
      FP6(a0, v2, v1, s2, s1); FP6(a1, v2, v1, s2, s1);
      FP6(a0, v1, v1, s2, s1); FP6(a1, v1, v1, s2, s1);
    } else if (MODE == 1) {  // all fp6 (no format change among the scaled ones)
      FP6(a0, v1, v2, s1, s2); FP6(a1, v1, v2, s1, s2);
      FP6(a0, v3, v2, s1, s2); FP6(a1, v3, v2, s1, s2);
      FP6(a0, v2, v1, s2, s1); FP6(a1, v2, v1, s2, s1);
      FP6(a0, v1, v1, s2, s1); FP6(a1, v1, v1, s2, s1);
    } else if (MODE == 2) {  // the scaled ones on accumulators of their own (no f16 -> scaled dependency)
      FP6(b0, v1, v2, s1, s2); FP6(b1, v1, v2, s1, s2);
      FP4(b0, v3, v2, s1, s2); FP4(b1, v3, v2, s1, s2);
      FP6(b0, v2, v1, s2, s1); FP6(b1, v2, v1, s2, s1);
      FP6(b0, v1, v1, s2, s1); FP6(b1, v1, v1, s2, s1);
    } else {                 // eighteen f16 MFMAs (the reference)
#pragma unroll
      for (int c = 0; c < 4; ++c) { F16(a0, h1, h2); F16(a1, h2, h1); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + b0[r] + b1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s + pad[0] * 0.f;
}
template <int MODE>
void run(const char *name) {
  const int blocks = 256, iters = 4000;
  float *out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  probe<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  probe<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-58s %.1f ns per step of 18 MFMAs\n", name, ms * 1e6 / iters);
}
int main() {
  run<3>("18 f16 MFMAs");
  run<0>("10 f16 + fp6, fp4, fp6, fp6 on the same accumulators");
  run<1>("10 f16 + 4 x fp6 on the same accumulators");
  run<2>("10 f16 + fp6, fp4, fp6, fp6 on accumulators of their own");
  return 0;
}
