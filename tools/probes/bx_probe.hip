// micro-probe: bf16 32x32x16 MFMA issue rate for the k_gmm_bx3 structure
//   MODE 0: one dependent chain, operands in registers
//   MODE 1: two independent chains
//   MODE 2: one chain, A operand from LDS (ds_read_b128), 3 reads per 6 MFMAs
//   DATA 0: zeros, DATA 1: pseudo-random bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MF(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)
__device__ unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
template <int MODE, int WPB>
__global__ __launch_bounds__(WPB * 64, 512 / (WPB * 64) >= 2 ? 2 : 1) void probe(float *out, int iters, int data) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2 * 15 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned seed = tid * 2654435761u + 12345u;
  for (int i = tid; i < 2 * 15 * 64; i += WPB * 64) {
    u32x4 v;
    for (int k = 0; k < 4; ++k) { seed = rnd(seed); v[k] = data ? ((seed & 0x0fff0fffu) | 0x30003000u) : 0u; }
    lds[i] = v;
  }
  __syncthreads();
  u32x4 b[15];
#pragma unroll
  for (int i = 0; i < 15; ++i)
    for (int k = 0; k < 4; ++k) { seed = rnd(seed); b[i][k] = data ? ((seed & 0x0fff0fffu) | 0x30003000u) : 0u; }
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 30; ++q) MF(b[(q + 1) % 15], b[q % 15], acc);
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 15; ++q) { MF(b[(q + 1) % 15], b[q], acc); MF(b[(q + 2) % 15], b[q], acc2); }
    } else {
      const u32x4 *cur = lds + (it & 1) * 15 * 64;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const u32x4 a1 = cur[(0 + c) * 64 + lane], a2 = cur[(5 + c) * 64 + lane], a3 = cur[(10 + c) * 64 + lane];
        MF(a3, b[c], acc); MF(a1, b[10 + c], acc); MF(a2, b[5 + c], acc);
        MF(a2, b[c], acc); MF(a1, b[5 + c], acc); MF(a1, b[c], acc);
      }
      if (MODE == 3) __syncthreads();
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
  out[blockIdx.x * WPB * 64 + tid] = s;
}
template <int MODE, int WPB>
void run(const char *name, int blocks, int iters, int data) {
  float *out; hipMalloc(&out, sizeof(float) * blocks * WPB * 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE, WPB><<<blocks, WPB * 64>>>(out, iters, data);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE, WPB><<<blocks, WPB * 64>>>(out, iters, data);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double mf = (double)blocks * WPB * iters * 30;
  printf("%-28s data %d blocks %4d x %d waves, iters %d: %.3f ms  %.0f TF/s  (%.1f cyc/MFMA/SIMD @2.4GHz; %.1f waves/SIMD)\n", name, data, blocks, WPB,
         iters, ms, mf * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024.0), blocks * WPB / 1024.0);
  hipFree(out);
}
int main() {
  for (int data = 0; data < 2; ++data) {
    run<0, 4>("1 chain regs", 512, 400, data);
    run<0, 4>("1 chain regs", 1024, 200, data);
    run<1, 4>("2 chains regs", 512, 400, data);
    run<0, 4>("1 chain regs (1 wave/SIMD)", 256, 800, data);
    run<1, 4>("2 chains regs (1 wave/SIMD)", 256, 800, data);
    run<2, 4>("A from LDS", 512, 400, data);
    run<3, 4>("A from LDS + barrier", 512, 400, data);
    run<2, 4>("A from LDS (4 waves/SIMD)", 1024, 200, data);
  }
  return 0;
}
