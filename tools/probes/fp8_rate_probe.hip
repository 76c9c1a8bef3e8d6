// micro-probe: issue rate of the 8-bit matrix instructions against f16, one wave per SIMD, two independent accumulators
// (the structure of k_gmm_fx2w's steps): v_mfma_f32_32x32x16_f16, v_mfma_f32_32x32x16_bf8_bf8 / _fp8_fp8 (the gfx940
// forms: 8 k-values per lane in 64 bits) and -- if the compiler has it -- v_mfma_f32_32x32x64_f8f6f4.
// Also checks the arithmetic of the bf8 form on small integers (exactly representable), so that the operand layout
// assumed by a kernel (lane = row / column index, 8 consecutive k per lane) can be trusted.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters) {
  extern __shared__ float pad[];  // > 80 KB: one workgroup per compute unit, one wave per SIMD
  const int lane = threadIdx.x & 63;
  f32x16 a0, a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  f16x8 h1, h2;
#pragma unroll
  for (int i = 0; i < 8; ++i) { h1[i] = (_Float16)(0.001f * (lane + i)); h2[i] = (_Float16)(0.002f * (lane - i)); }
  long b1 = 0x3838383838383838L + lane, b2 = 0x3434343434343434L + 3 * lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h2, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, h1, a1, 0, 0, 0);
      } else if (MODE == 1) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(b1, b2, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(b2, b1, a1, 0, 0, 0);
      } else {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b1, b2, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b2, b1, a1, 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s + pad[0] * 0.f;
}
// arithmetic check: A[i][k] = small integers as bf8, B[k][j] likewise: C = A B exactly
__global__ void check(float *C) {
  const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;   // lane holds row i (A) / column i (B), k = 8 kh .. 8 kh + 7
  unsigned long long a = 0, b = 0;
  for (int u = 0; u < 8; ++u) {
    const int k = 8 * kh + u;
    const float av = (float)((i + k) % 5 - 2), bv = (float)((2 * i + 3 * k) % 7 - 3);
    // bf8 = e5m2: the top byte of the f16 encoding of a value with <= 2 mantissa bits
    _Float16 ah = (_Float16)av, bh = (_Float16)bv;
    unsigned short ab, bb;
    memcpy(&ab, &ah, 2); memcpy(&bb, &bh, 2);
    a |= (unsigned long long)(ab >> 8) << (8 * u);
    b |= (unsigned long long)(bb >> 8) << (8 * u);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8((long)a, (long)b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = c[r];   // C layout of the 32x32 MFMAs: row = (r&3) + 8 (r>>2) + 4 (lane>>5), column = lane & 31
}
template <int MODE>
void run(const char *name) {
  const int blocks = 256, iters = 4000;
  float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = (double)iters * 32;   // MFMAs per wave
  printf("%-34s %.3f ms: %.2f ns per MFMA per SIMD, %.0f TFLOP/s on 256 CUs\n", name, ms, ms * 1e6 / n, 256.0 * 4 * n * 32768 / ms / 1e9);
  hipFree(out);
}
int main() {
  run<0>("v_mfma_f32_32x32x16_f16");
  run<1>("v_mfma_f32_32x32x16_bf8_bf8");
  run<2>("v_mfma_f32_32x32x16_fp8_fp8");
  float *C; hipMalloc(&C, 4096);
  check<<<1, 64>>>(C);
  float h[1024]; hipMemcpy(h, C, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float w = 0.f;
    for (int k = 0; k < 16; ++k) w += (float)((i + k) % 5 - 2) * (float)((2 * j + 3 * k) % 7 - 3);
    if (h[i * 32 + j] != w) ++bad;
  }
  printf("bf8 MFMA arithmetic / layout check: %d of 1024 elements differ\n", bad);
  return 0;
}
