// micro-probe: v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands -- operand layout, scale semantics and issue rate
// against v_mfma_f32_32x32x16_f16 (one wave per SIMD, two independent accumulators: the structure of k_gmm_fx2w's steps).
// The layout hypotheses are built on the host so that several can be tried by one binary:
//   H1: lane l holds row (A) / column (B) l & 31, K values 32 (l >> 5) .. + 31, value u in bits [6u, 6u + 6) of the
//       lane's 192-bit operand (registers 0 .. 5, little endian); the lane's scale byte applies to its 32 values
//   H2: the same with K = 16 (l >> 5) + (u & 15) + 32 (u >> 4)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters) {
  extern __shared__ float pad[];
  const int lane = threadIdx.x & 63;
  f32x16 a0, a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  f16x8 h1, h2;
#pragma unroll
  for (int i = 0; i < 8; ++i) { h1[i] = (_Float16)(0.001f * (lane + i)); h2[i] = (_Float16)(0.002f * (lane - i)); }
  i32x8 v1, v2;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v1[i] = 0x08208208 + lane * (i + 1); v2[i] = 0x04104104 + 3 * lane * (i + 2); }
  if (MODE == 1 || MODE == 3) { v1[6] = v1[7] = v2[6] = v2[7] = 0; }
  i32x8 v3;
#pragma unroll
  for (int i = 0; i < 8; ++i) v3[i] = i < 6 ? 0x0c30c30c + 5 * lane * (i + 1) : 0;
  const int s1 = 120 + (lane & 7), s2 = 125 - (lane & 3);
  float e0 = 0.f, e1 = 0.f, u0 = 0.f, u1 = 0.f, t0 = -0.01f * lane, t1 = -0.02f * lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h2, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, h1, a1, 0, 0, 0);
      } else if (MODE == 1) {  // fp6 e2m3 both
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v1, v2, a0, 2, 2, 0, s1, 0, s2);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v2, v1, a1, 2, 2, 0, s2, 0, s1);
      } else if (MODE == 2) {  // fp8 e4m3 both
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v1, v2, a0, 0, 0, 0, s1, 0, s2);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v2, v1, a1, 0, 0, 0, s2, 0, s1);
      } else if (MODE == 3) {  // one f16 then one fp6 alternating: the mix a correction class would issue (5 : 3)
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h2, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v2, v1, a1, 2, 2, 0, s2, 0, s1);
      } else if (MODE == 5) {  // fp6 with the B operand read from accumulation registers (k_gmm_fx2w's frames' side)
        asm volatile("" : "+a"(v2), "+a"(v1));
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v3, v2, a0, 2, 2, 0, s1, 0, s2);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v3, v1, a1, 2, 2, 0, s2, 0, s1);
      } else if (MODE == 6) {  // the F6 step's mix: 20 f16 MFMAs then 12 scaled ones, B from accumulation registers
        asm volatile("" : "+a"(v2), "+a"(v1));
        if (q < 10) {
          if (q & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, h1, a1, 0, 0, 0);
          else a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h2, a0, 0, 0, 0);
          if (q & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, h1, a1, 0, 0, 0);
          else a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h2, a0, 0, 0, 0);
        } else {
          a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v3, v2, a0, 2, 2, 0, s1, 0, s2);
          a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v3, v1, a1, 2, 2, 0, s2, 0, s1);
        }
      } else if (MODE == 7 || MODE == 8) {  // two exponentials + two additions behind every MFMA: f16 (7) / scaled fp6 (8)
        if (MODE == 7) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h2, a0, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v3, v2, a0, 2, 2, 0, s1, 0, s2);
        asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=v"(e0), "=v"(e1) : "v"(t0), "v"(t1));
        asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(u0), "+v"(u1) : "v"(t0), "v"(t1));
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 7) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, h1, a1, 0, 0, 0);
        else a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v3, v1, a1, 2, 2, 0, s2, 0, s1);
        asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=v"(e0), "=v"(e1) : "v"(t0), "v"(t1));
        asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(u0), "+v"(u1) : "v"(t0), "v"(t1));
        __builtin_amdgcn_sched_barrier(0);
      } else {                 // fp4
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v1, v2, a0, 4, 4, 0, s1, 0, s2);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v2, v1, a1, 4, 4, 0, s2, 0, s1);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s + pad[0] * 0.f + e0 + e1 + u0 + u1;
}

__global__ void check(const int *A, const int *B, const int *SA, const int *SB, float *C) {
  const int lane = threadIdx.x & 63;
  i32x8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = r < 6 ? A[lane * 6 + r] : 0; b[r] = r < 6 ? B[lane * 6 + r] : 0; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, SA[lane], 0, SB[lane]);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}

static float e2m3(int code) {  // sign, 2 exponent bits (bias 1), 3 mantissa bits; no infinities / NaNs
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -v : v;
}

template <int MODE>
void run(const char *name, int kper) {
  const int blocks = 256, iters = 3000;
  float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = (double)iters * 32;
  printf("%-44s %.3f ms: %.2f ns per MFMA per SIMD (%.2f ns per 16 K)\n", name, ms, ms * 1e6 / n, ms * 1e6 / n * 16.0 / kper);
  hipFree(out);
}

int main() {
  run<0>("v_mfma_f32_32x32x16_f16", 16);
  run<1>("v_mfma_scale_f32_32x32x64_f8f6f4 fp6 e2m3", 64);
  run<2>("v_mfma_scale_f32_32x32x64_f8f6f4 fp8 e4m3", 64);
  run<4>("v_mfma_scale_f32_32x32x64_f8f6f4 fp4", 64);
  run<3>("alternating f16 K=16 / fp6 K=64 (per pair /2)", 40);
  run<5>("fp6 e2m3, B from accumulation registers", 64);
  run<6>("F6 step mix (20 f16 + 12 scaled per 32), B from AGPRs", 40);
  run<7>("f16 K=16 + 2 v_exp + 2 v_add behind each", 16);
  run<8>("fp6 K=64 + 2 v_exp + 2 v_add behind each", 64);
  // layout check
  static int ca[32][64], cb[64][32], sa[32][2], sb[32][2];
  srand(7);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) { ca[i][k] = rand() & 63; cb[k][i] = rand() & 63; }
  for (int i = 0; i < 32; ++i) for (int h = 0; h < 2; ++h) { sa[i][h] = 120 + rand() % 12; sb[i][h] = 122 + rand() % 9; }
  int *dA, *dB, *dSA, *dSB; float *dC;
  hipMalloc(&dA, 64 * 6 * 4); hipMalloc(&dB, 64 * 6 * 4); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dC, 4096);
  for (int hyp = 1; hyp <= 2; ++hyp) {
    unsigned A[64][6], B[64][6]; int SA[64], SB[64];
    memset(A, 0, sizeof A); memset(B, 0, sizeof B);
    for (int l = 0; l < 64; ++l) {
      const int i = l & 31, kh = l >> 5;
      for (int u = 0; u < 32; ++u) {
        const int k = hyp == 1 ? 32 * kh + u : 16 * kh + (u & 15) + 32 * (u >> 4);
        const int bit = 6 * u;
        unsigned long long va = (unsigned long long)ca[i][k] << (bit & 31), vb = (unsigned long long)cb[k][i] << (bit & 31);
        A[l][bit >> 5] |= (unsigned)va; B[l][bit >> 5] |= (unsigned)vb;
        if ((bit >> 5) + 1 < 6) { A[l][(bit >> 5) + 1] |= (unsigned)(va >> 32); B[l][(bit >> 5) + 1] |= (unsigned)(vb >> 32); }
      }
      SA[l] = sa[i][kh] | 0x55aa00;  // junk in the other bytes: only byte 0 may count
      SB[l] = sb[i][kh] | 0x11000000;
    }
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA, 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB, 256, hipMemcpyHostToDevice);
    check<<<1, 64>>>(dA, dB, dSA, dSB, dC);
    float h[1024]; hipMemcpy(h, dC, 4096, hipMemcpyDeviceToHost);
    // expectation under "scale of K block kb = the scale of the lane that holds it": for H2 a lane's values straddle
    // both 32-blocks, so use the lane's scale per value
    int bad = 0; double worst = 0.0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double w = 0.0;
      for (int k = 0; k < 64; ++k) {
        const int kh = hyp == 1 ? k >> 5 : (k >> 4) & 1;
        w += (double)e2m3(ca[i][k]) * ldexp(1.0, sa[i][kh] - 127) * (double)e2m3(cb[k][j]) * ldexp(1.0, sb[j][kh] - 127);
      }
      const double d = fabs((double)h[i * 32 + j] - w);
      if (d > 1e-6 * fabs(w) + 1e-9) ++bad;
      if (d > worst) worst = d;
    }
    printf("fp6 layout hypothesis H%d: %d of 1024 elements differ (worst %.3g)\n", hyp, bad, worst);
  }
  return 0;
}
