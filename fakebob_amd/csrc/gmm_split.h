// gmm_split.h -- device helpers shared by the diagonal-GMM kernels (gmm_kernels.hip, gmm_wide_kernel.hip): the
// two-term f16 split of f32 operands and the online logsumexp in the log2 domain.
#pragma once
#include "fb_device.h"
#include "fb_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define FB_GMM_NEG (-3.0e38f)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fb_pow2f(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }

// v[8] -> two packed f16x8 fragments, v = f1 + f2 to 22 bits: f1 = f16(v) (round to nearest even), f2 = f16(v - f1) --
// the residual is exact in f32 (possibly a subnormal f16, which the matrix pipe keeps).  Four instructions per pair of
// values: v_cvt_pk_f16_f32, two v_fma_mix_f32 (x - f1 straight from the packed halves: no conversion back), v_cvt_pk_f16_f32.
__device__ __forceinline__ void fb_split2_frag(const float (&v)[8], u32x4 &f1, u32x4 &f2) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned a, b;
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(v[2 * i]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(v[2 * i + 1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(b) : "v"(r0), "v"(r1));
    f1[i] = a;
    f2[i] = b;
  }
}

// The frame operands of a lane of k_gmm_fx2 / k_gmm_fx2_sel / k_gsel_w (see the comment in k_gmm_fx2, gmm_kernels.hip): chunk c = dims 16c + 8h + i,
// two-term f16 splits of x (1.0 at position D) and x^2 under the load-time powers of two; returns the wave's range shift.
// (two steps, so that a kernel with several frames per lane can have every row's loads in flight before the first is used:
//  fb_fx_frame_load -- D a multiple of 4 -- and fb_fx_frame_make; fb_fx_frame_frags is both for one frame, any D)
template <int NK>
__device__ __forceinline__ void fb_fx_frame_load(const FbGmmDev &g, const float *__restrict__ feats, int row, int n_rows, int h,
                                                 float4 (&ft)[NK][2]) {
  const float *fr = feats + (size_t)(row < n_rows ? row : 0) * g.D;
#pragma unroll
  for (int c = 0; c < NK; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u) ft[c][u] = *reinterpret_cast<const float4 *>(fr + min(16 * c + 8 * h + 4 * u, g.D - 4));
}
template <int NK, bool PRELOADED>
__device__ __forceinline__ int fb_fx_frame_make(const FbGmmDev &g, const float *__restrict__ feats, const float4 (&ft)[NK][2], int row,
                                                int n_rows, int h, u32x4 (&bx1)[NK], u32x4 (&bx2)[NK], u32x4 (&bq1)[NK], u32x4 (&bq2)[NK]) {
  int sh = 0;
  {
    const bool ok = row < n_rows;
    const float *fr = feats + (size_t)(ok ? row : 0) * g.D;
    const float qs = fb_pow2f(g.kx2), xs = fb_pow2f(g.kx);  // exact power-of-two operand scalings
    float vv[NK][8], qq[NK][8];
    float amax = xs;
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      const int d0 = 16 * c + 8 * h;
      float *v = vv[c], *q = qq[c];
      if (PRELOADED || (g.D & 3) == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d = d0 + 4 * u;
          const float4 t = PRELOADED ? ft[c][u] : *reinterpret_cast<const float4 *>(fr + min(d, g.D - 4));
          const bool in = ok && d < g.D;
          v[4 * u + 0] = in ? t.x : 0.0f; v[4 * u + 1] = in ? t.y : 0.0f;
          v[4 * u + 2] = in ? t.z : 0.0f; v[4 * u + 3] = in ? t.w : 0.0f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (ok && d0 + i < g.D) ? fr[min(d0 + i, g.D - 1)] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __fmul_rn(__fmul_rn(v[i], v[i]), qs);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (d0 + i == g.D) ? xs : __fmul_rn(v[i], xs);
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fmaxf(fabsf(v[i]), q[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (amax >= 32768.0f) {  // wave-uniform; finite features only (the front-end produces nothing else)
      const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;  // amax in [2^ex, 2^(ex+1))
      sh = min(ex - 14, 100);
    }
    const float down = fb_pow2f(-sh);
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      if (sh) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { vv[c][i] = __fmul_rn(vv[c][i], down); qq[c][i] = __fmul_rn(qq[c][i], down); }
      }
      fb_split2_frag(vv[c], bx1[c], bx2[c]);
      fb_split2_frag(qq[c], bq1[c], bq2[c]);
    }
  }
  return sh;
}
template <int NK>
__device__ __forceinline__ int fb_fx_frame_frags(const FbGmmDev &g, const float *__restrict__ feats, int row, int n_rows, int h,
                                                 u32x4 (&bx1)[NK], u32x4 (&bx2)[NK], u32x4 (&bq1)[NK], u32x4 (&bq2)[NK]) {
  float4 none[NK][2];
  return fb_fx_frame_make<NK, false>(g, feats, none, row, n_rows, h, bx1, bx2, bq1, bq2);
}

#define FB_FX_MFMA(A, B, ACC) \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), ACC, 0, 0, 0)

// online logsumexp of k_gmm_fx2: fold 16 values into the state (m, s) at (stm, sts).
// The state lives in the log2 domain so that one value costs one (packed) fma, one v_exp_f32 and one (packed) add:
//   m = running maximum (a value, exact),  r = fl(m * L),  s = sum 2^(v * L - r),   L = fl(log2 e)
//   => logsumexp = ln2 * (r + log2 s); fb_lse_to_natural() converts to the (m, sum exp(v - m)) convention the
//   chunk merge / k_gmm_finalize use.  The reference point r only has to be the same for every term of s.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define FB_LOG2E_F 1.44269502162933349609375f  // fl(log2 e)
__device__ __forceinline__ void fb_lse_update16(const f32x16 &pv, float *__restrict__ stm, float *__restrict__ sts,
                                                float ls = FB_LOG2E_F) {  // ls = fl(log2 e) * 2^-kacc for scaled values
  float tm = FB_GMM_NEG;
#pragma unroll
  for (int r = 0; r < 16; ++r) tm = fmaxf(tm, pv[r]);
  const float m_old = *stm, s_old = *sts;
  const float m_new = fmaxf(m_old, tm);
  const float r_new = __fmul_rn(m_new, ls), r_old = __fmul_rn(m_old, ls);  // r_old = -inf at the start
  const f32x2 l2 = {ls, ls}, nr2 = {-r_new, -r_new};
  f32x2 acc = {s_old * __builtin_amdgcn_exp2f(r_old - r_new), 0.0f};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const f32x2 v2 = {pv[2 * r], pv[2 * r + 1]};
    const f32x2 t = __builtin_elementwise_fma(v2, l2, nr2);
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    acc += e;
  }
  *stm = m_new;
  *sts = acc[0] + acc[1];
}
// s (log2-domain state, see above) -> sum exp(v - m):  s * 2^(r - L m), evaluated in float64 (|r - L m| < 1e-4)
__device__ __forceinline__ float fb_lse_to_natural(float m, float s, float ls = FB_LOG2E_F) {
  const double d = (double)__fmul_rn(m, ls) - (double)ls * (double)m;
  return (float)((double)s * (1.0 + d * 0.6931471805599453 * (1.0 + d * 0.34657359027997264)));
}

