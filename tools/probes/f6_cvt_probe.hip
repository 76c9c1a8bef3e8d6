// micro-probe: v_cvt_scalef32_pk32_fp6_f16 (32 f16 values + one f32 scale -> 32 packed e2m3 codes): what the scale means,
// how it rounds and saturates, what it does with f16 subnormals, and that its packing order is the one
// v_mfma_scale_f32_32x32x64_f8f6f4 unpacks (element u in bits [6u, 6u + 6) of the 192-bit operand).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void conv(const _Float16 *in, unsigned *out, const float *sc) {
  f16x32 v;
  for (int i = 0; i < 32; ++i) v[i] = in[threadIdx.x * 32 + i];
  u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, sc[threadIdx.x]);
  for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
// C = A B with A host-packed codes (scale byte 127 = 1.0), B converted on the device from f16 with scale 2^e (and the
// MFMA given the matching scale byte): C must equal sum_k a_ik * fp6(b_kj / 2^e) * 2^e
__global__ void mm(const int *A, const _Float16 *Bf, int ebyte, float *C) {
  const int lane = threadIdx.x & 63;
  i32x8 a, b;
  for (int r = 0; r < 8; ++r) a[r] = r < 6 ? A[lane * 6 + r] : 0;
  f16x32 v;
  for (int u = 0; u < 32; ++u) v[u] = Bf[lane * 32 + u];
  const u32x6 q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, __uint_as_float((unsigned)ebyte << 23));
  for (int r = 0; r < 8; ++r) b[r] = r < 6 ? (int)q[r] : 0;
  asm volatile("" : "+a"(b));  // parked in accumulation registers: the MFMA must read it from there
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, 127, 0, ebyte);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}
static float e2m3(int code) {
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -v : v;
}
static float rne_e2m3(float x) {  // nearest e2m3 value, ties to the even code, saturating at 7.5
  const float a = fabsf(x);
  int code;
  if (a >= 7.5f) code = 31;
  else {
    const float step = a < 2.0f ? 0.125f : (a < 4.0f ? 0.25f : 0.5f);
    const float base = a < 2.0f ? 0.0f : (a < 4.0f ? 2.0f : 4.0f);
    const int cb = a < 2.0f ? 0 : (a < 4.0f ? 16 : 24);
    code = cb + (int)nearbyintf((a - base) / step);
  }
  return x < 0 ? -e2m3(code) : e2m3(code);
}
int main() {
  const float vals[32] = {0.f, 0.0625f, 0.07f, 0.125f, 0.1875f, 0.19f, 0.3125f, 1.0f, 1.06f, 1.0625f, 1.1875f, 1.9f, 1.97f, 2.1f, 2.125f,
                          3.9f, 4.2f, 4.25f, 7.5f, 7.7f, 7.75f, 8.f, 100.f, -0.0625f, -1.06f, -7.9f, -3.f, 0.01f, 5.96e-8f, 6.0e-6f, 3.0e-5f, 65504.f};
  const float scales[8] = {1.0f, 2.0f, 0.25f, 9.5367431640625e-07f /* 2^-20 */, 3.0f /* not a power of two */, 0.0f, 4096.0f, 1.5f};
  _Float16 in[8 * 32]; float sc[8];
  for (int t = 0; t < 8; ++t) { sc[t] = scales[t]; for (int i = 0; i < 32; ++i) in[t * 32 + i] = (_Float16)vals[i]; }
  _Float16 *din; unsigned *dout; float *dsc;
  hipMalloc(&din, sizeof in); hipMalloc(&dout, 8 * 6 * 4); hipMalloc(&dsc, sizeof sc);
  hipMemcpy(din, in, sizeof in, hipMemcpyHostToDevice); hipMemcpy(dsc, sc, sizeof sc, hipMemcpyHostToDevice);
  conv<<<1, 8>>>(din, dout, dsc);
  unsigned out[8 * 6]; hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
  for (int t = 0; t < 8; ++t) {
    int bad = 0;
    printf("scale %-12g:", scales[t]);
    for (int i = 0; i < 32; ++i) {
      const int bit = 6 * i;
      unsigned long long w = out[t * 6 + (bit >> 5)];
      if ((bit >> 5) + 1 < 6) w |= (unsigned long long)out[t * 6 + (bit >> 5) + 1] << 32;
      const int code = (int)((w >> (bit & 31)) & 63);
      const float got = e2m3(code);
      // model: the scale's EXPONENT only (a power of two), value / 2^e rounded to nearest even, saturating
      int ex; frexpf(scales[t], &ex);
      const float p2 = scales[t] > 0.f ? ldexpf(1.0f, ex - 1) : 1.0f;
      const float want = rne_e2m3((float)in[t * 32 + i] / p2);
      if (got != want) { ++bad; printf(" [%g -> %g, model %g]", (double)(float)in[t * 32 + i], got, want); }
    }
    printf(" %d of 32 differ from the model\n", bad);
  }
  // MFMA consistency
  static int ca[32][64]; static float bf[64][32];
  srand(11);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) { ca[i][k] = rand() & 63; bf[k][i] = (float)(_Float16)(((rand() % 2001) - 1000) * 1.0e-3f); }
  unsigned A[64][6]; _Float16 B[64][32];
  memset(A, 0, sizeof A);
  for (int l = 0; l < 64; ++l) for (int u = 0; u < 32; ++u) {
    const int i = l & 31, k = 32 * (l >> 5) + u, bit = 6 * u;
    const unsigned long long va = (unsigned long long)ca[i][k] << (bit & 31);
    A[l][bit >> 5] |= (unsigned)va;
    if ((bit >> 5) + 1 < 6) A[l][(bit >> 5) + 1] |= (unsigned)(va >> 32);
    B[l][u] = (_Float16)bf[k][i];
  }
  int *dA; _Float16 *dB; float *dC;
  hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
  const int ebyte = 127 - 3;  // 2^-3: |b| <= 1 -> |b / s| <= 8
  mm<<<1, 64>>>(dA, dB, ebyte, dC);
  float h[1024]; hipMemcpy(h, dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0; double worst = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double w = 0;
    for (int k = 0; k < 64; ++k) w += (double)e2m3(ca[i][k]) * (double)rne_e2m3(bf[k][j] * 8.0f) * 0.125;
    const double d = fabs(w - (double)h[i * 32 + j]);
    if (d > 1e-5 * fabs(w) + 1e-6) ++bad;
    if (d > worst) worst = d;
  }
  printf("device-converted B (from accumulation registers) against host-packed A: %d of 1024 differ (worst %.3g)\n", bad, worst);
  return 0;
}
