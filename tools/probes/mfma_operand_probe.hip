// Where may the operands of v_mfma_f32_32x32x16_f16 live at full rate?  (design data for the register plan of
// k_gmm_fx2w: accumulators in vector registers would spare the update its v_accvgpr_read copies, if the frame
// operands that do not fit beside them could be read from the accumulation file.)
// 64 MFMAs per trip on two alternating accumulators, one wave per SIMD (100 KB of LDS per workgroup):
//   C/D in {v, a}  x  B in {v, a}  (A always v);  plus the same with K vector instructions per gap.
// hipcc --offload-arch=gfx950 -O3 -o mfma_operand_probe mfma_operand_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MF(CD, BC) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+" CD(x0) : "v"(a), BC(b0)); \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+" CD(x1) : "v"(a), BC(b1));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63;
  f16x8 a, b0, b1;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * ((lane + i) & 15)); b0[i] = (_Float16)(0.02f * ((lane * 3 + i) & 7)); b1[i] = (_Float16)(0.03f * ((lane + i) & 3)); }
  f32x16 x0, x1, y;
  for (int i = 0; i < 16; ++i) { x0[i] = 0.f; x1[i] = 0.f; y[i] = -1.0f * i - 0.01f * lane; }
  float r[8], c0 = 0.999f, c1 = -1e-3f;
  for (int i = 0; i < 8; ++i) r[i] = -0.001f * (lane + i);
  if (MODE == 20 || MODE == 21) asm volatile("" : "+a"(y));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int gq = 0; gq < 32; ++gq) {
      if (MODE == 0) { MF("v", "v") }
      else if (MODE == 1) { MF("v", "a") }
      else if (MODE == 2) { MF("a", "v") }
      else if (MODE == 3) { MF("a", "a") }
      else if (MODE == 10 || MODE == 11) {  // VALU-bound gap, values already in vector registers: 2 x (fma, exp, add) + 2 fma
        if (MODE == 10) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(b0)); }
        else { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(x0) : "v"(a), "v"(b0)); }
        asm volatile("v_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_add_f32 %4, %4, %6\nv_add_f32 %5, %5, %7"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "v"(r[6]), "v"(r[7]), "v"(c0), "v"(c1));
        if (MODE == 10) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x1) : "v"(a), "v"(b1)); }
        else { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(x1) : "v"(a), "v"(b1)); }
        asm volatile("v_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_add_f32 %4, %4, %6\nv_add_f32 %5, %5, %7"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "v"(r[6]), "v"(r[7]), "v"(c0), "v"(c1));
      } else if (MODE == 20 || MODE == 21) {  // the same gap fed from the accumulation file: + 2 v_accvgpr_read (20), + max3 too (21)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(x0) : "v"(a), "v"(b0));
        asm volatile("v_accvgpr_read_b32 %6, %10\nv_accvgpr_read_b32 %7, %10\nv_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_add_f32 %4, %4, %6\nv_add_f32 %5, %5, %7"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1), "a"(y[0]));
        if (MODE == 21) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[4]) : "v"(r[6]), "v"(r[7]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(x1) : "v"(a), "v"(b1));
        asm volatile("v_accvgpr_read_b32 %6, %10\nv_accvgpr_read_b32 %7, %10\nv_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_add_f32 %4, %4, %6\nv_add_f32 %5, %5, %7"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1), "a"(y[1]));
        if (MODE == 21) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[5]) : "v"(r[6]), "v"(r[7]));
      } else if (MODE == 30 || MODE == 31) {  // 3 values per gap (the delta items' density with 10 gaps per update half)
        if (MODE == 30) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(b0)); }
        else { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "a"(b0)); }
        asm volatile("v_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_fma_f32 %2, %7, %8, %9\nv_exp_f32 %3, %3\nv_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_add_f32 %6, %6, %0\nv_add_f32 %7, %7, %1\nv_add_f32 %6, %6, %2"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1));
        if (MODE == 30) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x1) : "v"(a), "v"(b1)); }
        else { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(x1) : "v"(a), "a"(b1)); }
        asm volatile("v_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_fma_f32 %2, %7, %8, %9\nv_exp_f32 %3, %3\nv_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_add_f32 %6, %6, %0\nv_add_f32 %7, %7, %1\nv_add_f32 %6, %6, %2"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += r[i];
  for (int i = 0; i < 16; ++i) s += x0[i] + x1[i] + y[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + sm[threadIdx.x];
}

template <int MODE>
void run(const char *name) {
  static float *out = nullptr;
  if (!out) hipMalloc(&out, sizeof(float) * 256 * 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 4000;
  probe<MODE><<<256, 256, 100 * 1024>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<256, 256, 100 * 1024>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-64s %8.3f ms  %7.2f ns per MFMA\n", name, ms, ms * 1e6 / (iters * 64.0));
}
int main() {
  for (int w = 0; w < 20; ++w) run<0>("(warm-up)");
  run<0>("C/D v, B v"); run<1>("C/D v, B a"); run<2>("C/D a, B v"); run<3>("C/D a, B a");
  run<10>("C/D v, gap: 2 fma 2 exp 2 add"); run<11>("C/D a, gap: 2 fma 2 exp 2 add");
  run<20>("C/D a, gap: 2 accvgpr_read + 2 fma 2 exp 2 add"); run<21>("C/D a, gap: 2 accvgpr_read + max3 + 2 fma 2 exp 2 add");
  run<30>("C/D v, B v, gap: 3 fma 3 exp 3 add"); run<31>("C/D v, B a, gap: 3 fma 3 exp 3 add");
  run<0>("C/D v, B v (again)");
  return 0;
}
