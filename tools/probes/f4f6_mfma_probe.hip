// micro-probe: v_mfma_scale_f32_32x32x64_f8f6f4 with MIXED operand formats -- A in fp4 (e2m1: 32 codes in 128 bits, four
// registers), B in fp6 (e2m3: 192 bits, six registers) --: arithmetic / packing check on random codes and the issue rate
// against the fp6 x fp6 form.  (k_gmm_fx2w's F6 class takes the low term of a parameter split in fp4.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void check(const int *A, const int *B, const int *SA, const int *SB, float *C) {
  const int lane = threadIdx.x & 63;
  i32x8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = r < 4 ? A[lane * 4 + r] : 0; b[r] = r < 6 ? B[lane * 6 + r] : 0; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 2, 0, SA[lane], 0, SB[lane]);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void rate(float *out, int iters) {
  extern __shared__ float pad[];
  const int lane = threadIdx.x & 63;
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  i32x8 v1, v2;
  for (int i = 0; i < 8; ++i) { v1[i] = i < 6 ? 0x08208208 + lane * (i + 1) : 0; v2[i] = i < 6 ? 0x04104104 + 3 * lane * (i + 2) : 0; }
  const int s1 = 120 + (lane & 7), s2 = 125 - (lane & 3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v1, v2, a0, 2, 2, 0, s1, 0, s2);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v2, v1, a1, 2, 2, 0, s2, 0, s1);
      } else {
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v1, v2, a0, 4, 2, 0, s1, 0, s2);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v2, v1, a1, 4, 2, 0, s2, 0, s1);
      }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s + pad[0] * 0.f;
}
static float e2m3(int code) {
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -v : v;
}
static float e2m1(int code) {  // sign, 2 exponent bits (bias 1), 1 mantissa bit: 0, .5, 1, 1.5, 2, 3, 4, 6
  const int s = code >> 3, e = (code >> 1) & 3, m = code & 1;
  const float v = e == 0 ? m * 0.5f : ldexpf(1.0f + m * 0.5f, e - 1);
  return s ? -v : v;
}
template <int MODE>
void run(const char *name) {
  const int blocks = 256, iters = 3000;
  float *out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rate<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  rate<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  rate<MODE><<<blocks, 256, 90 * 1024>>>(out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.2f ns per MFMA per SIMD\n", name, ms * 1e6 / ((double)iters * 32));
}
int main() {
  run<0>("fp6 x fp6");
  run<1>("fp4 (A) x fp6 (B)");
  static int ca[32][64], cb[64][32], sa[32][2], sb[32][2];
  srand(5);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) { ca[i][k] = rand() & 15; cb[k][i] = rand() & 63; }
  for (int i = 0; i < 32; ++i) for (int h = 0; h < 2; ++h) { sa[i][h] = 120 + rand() % 12; sb[i][h] = 122 + rand() % 9; }
  unsigned A[64][4], B[64][6]; int SA[64], SB[64];
  memset(A, 0, sizeof A); memset(B, 0, sizeof B);
  for (int l = 0; l < 64; ++l) {
    const int i = l & 31, kh = l >> 5;
    for (int u = 0; u < 32; ++u) {
      const int k = 32 * kh + u;
      A[l][(4 * u) >> 5] |= (unsigned)ca[i][k] << ((4 * u) & 31);
      const int bit = 6 * u;
      const unsigned long long vb = (unsigned long long)cb[k][i] << (bit & 31);
      B[l][bit >> 5] |= (unsigned)vb;
      if ((bit >> 5) + 1 < 6) B[l][(bit >> 5) + 1] |= (unsigned)(vb >> 32);
    }
    SA[l] = sa[i][kh]; SB[l] = sb[i][kh];
  }
  int *dA, *dB, *dSA, *dSB; float *dC;
  (void)hipMalloc(&dA, sizeof A); (void)hipMalloc(&dB, sizeof B); (void)hipMalloc(&dSA, 256); (void)hipMalloc(&dSB, 256); (void)hipMalloc(&dC, 4096);
  (void)hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
  (void)hipMemcpy(dSA, SA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dSB, SB, 256, hipMemcpyHostToDevice);
  check<<<1, 64>>>(dA, dB, dSA, dSB, dC);
  float h[1024]; (void)hipMemcpy(h, dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0; double worst = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double w = 0;
    for (int k = 0; k < 64; ++k) w += (double)e2m1(ca[i][k]) * ldexp(1.0, sa[i][k >> 5] - 127) * (double)e2m3(cb[k][j]) * ldexp(1.0, sb[j][k >> 5] - 127);
    const double d = fabs(w - (double)h[i * 32 + j]);
    if (d > 1e-6 * fabs(w) + 1e-9) ++bad;
    if (d > worst) worst = d;
  }
  printf("fp4 x fp6: %d of 1024 elements differ (worst %.3g)\n", bad, worst);
  return 0;
}
