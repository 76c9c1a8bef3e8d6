// Issue cost of the vector instructions the logsumexp update is made of, alone and threaded between f16 MFMAs,
// one wave per SIMD (100 KB of LDS requested per workgroup).  ns per instruction; v_fma_f32 = 4 cycles is the yardstick.
// hipcc --offload-arch=gfx950 -O3 -o valu_cost_probe valu_cost_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
// 8 instructions on 8 independent registers
#define I8(OP) asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(c0), "v"(c1));
#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %12, %13\n"
#define OP_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %12, %13\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %" #i ", %12\n"
#define OP_NOP(i) "s_nop 0\n"
#define OP_EXPNOP(i) "v_exp_f32 %" #i ", %" #i "\ns_nop 0\n"
#define OP_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %12\n"
#define OP_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n"
#define OP_CVT(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_EXP16(i) "v_exp_f16 %" #i ", %" #i "\n"
// packed ops on the 4 register pairs d[0..3] (64-bit each)
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define OP_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %" #i "\n"
#define OP_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %" #i "\n"
#define P4(OP) asm volatile(OP(8) OP(9) OP(10) OP(11) OP(8) OP(9) OP(10) OP(11) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(c0), "v"(c1));

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63;
  float r[8]; f32x2 d[4];
  for (int i = 0; i < 8; ++i) r[i] = -0.001f * (lane + i);
  for (int i = 0; i < 4; ++i) d[i] = f32x2{0.5f + 0.001f * lane, 0.25f};
  float c0 = 0.999f, c1 = -1e-3f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * ((lane + i) & 15)); b[i] = (_Float16)(0.02f * ((lane * 3 + i) & 7)); }
  f32x16 x0, x1, y;
  for (int i = 0; i < 16; ++i) { x0[i] = 0.f; x1[i] = 0.f; y[i] = 1.0f * i; }
  if (MODE == 13) asm volatile("" : "+a"(y));
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { REP8(I8(OP_FMA)) }
    else if (MODE == 1) { REP8(I8(OP_EXP)) }
    else if (MODE == 2) { REP8(I8(OP_MAX3)) }
    else if (MODE == 3) { REP8(I8(OP_ADD)) }
    else if (MODE == 4) { REP8(I8(OP_NOP)) }
    else if (MODE == 5) { REP8(I8(OP_EXPNOP)) }
    else if (MODE == 6) { REP8(P4(OP_PKFMA)) }
    else if (MODE == 7) { REP8(P4(OP_PKADD)) }
    else if (MODE == 8) { REP8(I8(OP_LDEXP)) }
    else if (MODE == 9) { REP8(I8(OP_FRACT)) }
    else if (MODE == 10) { REP8(I8(OP_CVT)) }
    else if (MODE == 11) { REP8(I8(OP_RCP)) }
    else if (MODE == 12) { REP8(I8(OP_EXP16)) }
    else if (MODE == 13) {  // 64 v_accvgpr_read
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(y[i])); r[i & 7] = t; }
      }
    } else if (MODE >= 20 && MODE < 40) {
      // 64 gaps: one MFMA (two alternating accumulators) + K instructions; 20 + K: K x v_fma_f32 ; 30 + K: K x v_exp_f32
#pragma unroll
      for (int gq = 0; gq < 64; ++gq) {
        if (gq & 1) x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x1, 0, 0, 0);
        else x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x0, 0, 0, 0);
        constexpr int K = MODE >= 30 ? MODE - 30 : MODE - 20;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (MODE >= 30) asm volatile("v_exp_f32 %0, %0" : "+v"(r[k & 7]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k & 7]) : "v"(c0), "v"(c1));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE >= 40 && MODE < 60) {
      // the update's gap: MFMA + pk_fma + 2 exp (+ s_nop) + pk_add ; 41: the add one gap late (no s_nop) ; 42: + 2 accvgpr_read
#pragma unroll
      for (int gq = 0; gq < 64; ++gq) {
        if (gq & 1) x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x1, 0, 0, 0);
        else x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x0, 0, 0, 0);
        if (MODE == 40)
          asm volatile("v_pk_fma_f32 %0, %0, %2, %3\nv_exp_f32 %4, %4\nv_exp_f32 %1, %1\ns_nop 0\nv_pk_add_f32 %2, %2, %0" : "+v"(d[0]), "+v"(r[0]), "+v"(d[1]) : "v"(d[2]), "v"(r[1]));
        else if (MODE == 41)
          asm volatile("v_pk_add_f32 %2, %2, %0\nv_pk_fma_f32 %0, %0, %2, %3\nv_exp_f32 %4, %4\nv_exp_f32 %1, %1" : "+v"(d[0]), "+v"(r[0]), "+v"(d[1]) : "v"(d[2]), "v"(r[1]));
        else if (MODE == 42)
          asm volatile("v_accvgpr_read_b32 %1, %4\nv_accvgpr_read_b32 %5, %4\nv_pk_add_f32 %2, %2, %0\nv_pk_fma_f32 %0, %0, %2, %3\nv_exp_f32 %5, %5\nv_exp_f32 %1, %1" : "+v"(d[0]), "+v"(r[0]), "+v"(d[1]) : "v"(d[2]), "a"(y[0]), "v"(r[1]));
        else if (MODE == 43) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[0]) : "v"(d[2]), "v"(d[3]));
        else if (MODE == 44) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\nv_pk_fma_f32 %1, %1, %2, %3" : "+v"(d[0]), "+v"(d[1]) : "v"(d[2]), "v"(d[3]));
        else if (MODE == 45) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[0]) : "v"(d[2]));
        else if (MODE == 46)  // the unpacked form of the update gap
          asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_exp_f32 %0, %0\nv_exp_f32 %1, %1\ns_nop 0\nv_add_f32 %2, %2, %0\nv_add_f32 %3, %3, %1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(c0), "v"(c1));
        else if (MODE == 47) asm volatile("v_pk_fma_f32 %0, %0, %3, %4\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2" : "+v"(d[0]), "+v"(r[0]), "+v"(r[1]) : "v"(d[2]), "v"(d[3]));
        else if (MODE == 48) asm volatile("v_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_pk_add_f32 %0, %0, %3" : "+v"(d[0]), "+v"(r[0]), "+v"(r[1]) : "v"(d[2]));
        else if (MODE == 49)  // independent pk_fma (no chain through the gap's own result)
          asm volatile("v_pk_fma_f32 %0, %3, %3, %4\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_pk_add_f32 %5, %5, %3" : "=v"(d[0]), "+v"(r[0]), "+v"(r[1]) : "v"(d[2]), "v"(d[3]), "v"(d[1]));
        else if (MODE == 50)  // the update's gap with nothing consumed in the gap that produces it
          asm volatile("v_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_add_f32 %4, %4, %6\nv_add_f32 %5, %5, %7"
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "v"(r[6]), "v"(r[7]), "v"(c0), "v"(c1));
        else if (MODE == 51)  // + two register copies and a max3
          asm volatile("v_accvgpr_read_b32 %6, %10\nv_accvgpr_read_b32 %7, %10\nv_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_add_f32 %4, %4, %6\nv_add_f32 %5, %5, %7\nv_max3_f32 %4, %4, %8, %9"
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c0), "v"(c1), "a"(y[0]));
        else if (MODE == 52)
          asm volatile("v_accvgpr_read_b32 %0, %6\nv_accvgpr_read_b32 %1, %6\nv_accvgpr_read_b32 %2, %6\nv_accvgpr_read_b32 %3, %6\nv_max3_f32 %4, %4, %0, %1\nv_max3_f32 %5, %5, %2, %3"
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "a"(y[0]));
        else if (MODE == 53)
          asm volatile("v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_add_f32 %2, %2, %4\nv_add_f32 %3, %3, %4" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(c0));
        else if (MODE == 54)  // 50 with the exponentials first
          asm volatile("v_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_add_f32 %4, %4, %6\nv_add_f32 %5, %5, %7"
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "v"(r[6]), "v"(r[7]), "v"(c0), "v"(c1));
        else if (MODE == 55)  // the kernel's present gap: fma -> exp of the SAME value, add of the previous one
          asm volatile("v_fma_f32 %0, %6, %8, %9\nv_fma_f32 %1, %7, %8, %9\nv_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_add_f32 %4, %4, %2\nv_add_f32 %5, %5, %3"
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "v"(r[6]), "v"(r[7]), "v"(c0), "v"(c1));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += r[i];
  for (int i = 0; i < 4; ++i) s += d[i][0] + d[i][1];
  for (int i = 0; i < 16; ++i) s += x0[i] + x1[i] + y[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + sm[threadIdx.x];
}

template <int MODE>
void run(const char *name, double per_iter) {
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
  printf("%-52s %8.3f ms  %7.2f ns per unit\n", name, ms, ms * 1e6 / (iters * per_iter));
}
int main() {
  for (int w = 0; w < 20; ++w) run<20>("(warm-up)", 64);
  run<0>("v_fma_f32", 64); run<1>("v_exp_f32", 64); run<2>("v_max3_f32", 64); run<3>("v_add_f32", 64);
  run<4>("s_nop 0", 64); run<5>("v_exp_f32 + s_nop 0", 64); run<6>("v_pk_fma_f32", 64); run<7>("v_pk_add_f32", 64);
  run<8>("v_ldexp_f32", 64); run<9>("v_fract_f32", 64); run<10>("v_cvt_i32_f32", 64); run<11>("v_rcp_f32", 64);
  run<12>("v_exp_f16", 64); run<13>("v_accvgpr_read_b32", 64);
  run<20>("gap: MFMA + 0 fma", 64); run<22>("gap: MFMA + 2 fma", 64); run<24>("gap: MFMA + 4 fma", 64);
  run<26>("gap: MFMA + 6 fma", 64); run<27>("gap: MFMA + 7 fma", 64); run<28>("gap: MFMA + 8 fma", 64);
  run<29>("gap: MFMA + 9 fma", 64);
  run<31>("gap: MFMA + 1 exp", 64); run<32>("gap: MFMA + 2 exp", 64); run<33>("gap: MFMA + 3 exp", 64);
  run<40>("gap: MFMA + pk_fma, 2 exp, nop, pk_add", 64); run<41>("gap: MFMA + pk_add(prev), pk_fma, 2 exp", 64);
  run<42>("gap: as 41 + 2 accvgpr_read", 64);
  run<43>("gap: MFMA + 1 pk_fma", 64); run<44>("gap: MFMA + 2 pk_fma", 64); run<45>("gap: MFMA + 1 pk_add", 64);
  run<46>("gap: MFMA + 2 fma, 2 exp, nop, 2 add (unpacked)", 64); run<47>("gap: MFMA + pk_fma, 2 exp", 64);
  run<48>("gap: MFMA + 2 exp, pk_add", 64); run<49>("gap: MFMA + pk_fma(indep), 2 exp, pk_add(indep)", 64);
  run<50>("gap: MFMA + 2 fma, 2 exp, 2 add, independent", 64); run<54>("gap: MFMA + 2 exp, 2 fma, 2 add, independent", 64);
  run<55>("gap: MFMA + 2 fma -> 2 exp (same value), 2 add(prev)", 64);
  run<51>("gap: as 50 + 2 accvgpr_read + max3", 64); run<52>("gap: MFMA + 4 accvgpr_read + 2 max3", 64);
  run<53>("gap: MFMA + 2 exp, 2 add", 64);
  run<20>("gap: MFMA + 0 fma (again)", 64); run<0>("v_fma_f32 (again)", 64); run<1>("v_exp_f32 (again)", 64);
  return 0;
}
