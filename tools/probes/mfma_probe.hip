// micro-probe: f32 MFMA issue rate under the k_gmm structure (chained accumulators, A operand from LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2464 * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2464 * 2; i += 256) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  float xf[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) xf[i] = 0.01f * (lane + i);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float *prow = lds + (lane & 31) * 76 + (lane >> 5) * 36;
  for (int it = 0; it < iters; ++it) {
    const float *pr = prow + (it & 1) * 2464;
    if (MODE == 0) {  // registers only
#pragma unroll
      for (int q = 0; q < 36; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[(q + 1) % 36], xf[q], acc, 0, 0, 0);
    } else {  // A operand from LDS (ds_read_b128)
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float4 p = *reinterpret_cast<const float4 *>(pr + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.x, xf[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.y, xf[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.z, xf[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.w, xf[4 * q + 3], acc, 0, 0, 0);
      }
      if (MODE == 2) __syncthreads();
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
void run(const char *name, int blocks, int iters) {
  float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double mf = (double)blocks * 4 * iters * 36;
  printf("%-22s blocks %4d iters %d: %.3f ms  %.1f TF/s  (%.1f cycles/MFMA/SIMD @2.4GHz, %d waves/SIMD)\n", name, blocks, iters, ms,
         mf * 4096 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024.0), blocks * 4 / 1024);
  hipFree(out);
}
int main() {
  for (int b : {1024, 2048, 4096}) {
    run<0>("regs only", b, 56 * 8 / (b / 1024));
    run<1>("A from LDS", b, 56 * 8 / (b / 1024));
    run<2>("A from LDS + barrier", b, 56 * 8 / (b / 1024));
  }
  return 0;
}
