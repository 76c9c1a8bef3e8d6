// fb_device.h -- device-side helpers shared by the gfx950 kernels.
//
// RNG contract (identical, op for op, to the CPU oracle so that noise -- and
// therefore every int16 sample and every attack decision -- is bit-identical):
//   Philox4x32-10, key = seed, counter = (n/4, pair j, iteration, stream);
//   the 4 output words give 4 consecutive samples via two float32 Box-Muller
//   transforms built only from exactly-rounded primitives.
// The library is compiled with -ffp-contract=off; every fused op is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FB_WAVE 64

// Memory order of every word that crosses workgroups inside a launch (agent-scope atomic stores and
// loads whose "ready" travels in the data word itself, no fences -- gfx950 hardware behaviour, not what the HIP memory model
// promises for neighbouring plain stores).  -DFB_FENCED builds the same kernels with release stores / acquire loads /
// acq_rel read-modify-writes at agent scope -- the compiler then writes back / invalidates the L2 path around each of them,
// which is what the memory model asks for and what costs microseconds per workgroup on the eight-XCD part.  Both builds
// must give the same bits (tests/test_gpu_fenced.py: fakebob_amd/lib/libfakebob_hip_fenced.so): a race in the fence-free
// protocol would show as a difference there.
#ifdef FB_FENCED
#define FB_XCH_ST __ATOMIC_RELEASE
#define FB_XCH_LD __ATOMIC_ACQUIRE
#define FB_XCH_RMW __ATOMIC_ACQ_REL
#else
#define FB_XCH_ST __ATOMIC_RELAXED
#define FB_XCH_LD __ATOMIC_RELAXED
#define FB_XCH_RMW __ATOMIC_RELAXED
#endif

__device__ __forceinline__ void fb_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float fb_ln_u(float u) {  // u in (0,1]
  uint32_t bits = __float_as_uint(u);
  int e = (int)(bits >> 23) - 127;
  float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
  if (m > 1.41421354f) { m = __fmul_rn(m, 0.5f); e += 1; }
  float t = __fsub_rn(m, 1.0f);
  float p = -1.0f / 20.0f;
  p = __fmaf_rn(p, t, 1.0f / 19.0f);
  p = __fmaf_rn(p, t, -1.0f / 18.0f);
  p = __fmaf_rn(p, t, 1.0f / 17.0f);
  p = __fmaf_rn(p, t, -1.0f / 16.0f);
  p = __fmaf_rn(p, t, 1.0f / 15.0f);
  p = __fmaf_rn(p, t, -1.0f / 14.0f);
  p = __fmaf_rn(p, t, 1.0f / 13.0f);
  p = __fmaf_rn(p, t, -1.0f / 12.0f);
  p = __fmaf_rn(p, t, 1.0f / 11.0f);
  p = __fmaf_rn(p, t, -1.0f / 10.0f);
  p = __fmaf_rn(p, t, 1.0f / 9.0f);
  p = __fmaf_rn(p, t, -1.0f / 8.0f);
  p = __fmaf_rn(p, t, 1.0f / 7.0f);
  p = __fmaf_rn(p, t, -1.0f / 6.0f);
  p = __fmaf_rn(p, t, 1.0f / 5.0f);
  p = __fmaf_rn(p, t, -1.0f / 4.0f);
  p = __fmaf_rn(p, t, 1.0f / 3.0f);
  p = __fmaf_rn(p, t, -1.0f / 2.0f);
  p = __fmaf_rn(p, t, 1.0f);
  p = __fmul_rn(p, t);
  return __fmaf_rn((float)e, 0.693147182f, p);
}

__device__ __forceinline__ void fb_sincos_q(float a, float &s, float &c) {  // a in [0, pi/2)
  float a2 = __fmul_rn(a, a);
  float ps = -1.0f / 1307674368000.0f;
  ps = __fmaf_rn(ps, a2, 1.0f / 6227020800.0f);
  ps = __fmaf_rn(ps, a2, -1.0f / 39916800.0f);
  ps = __fmaf_rn(ps, a2, 1.0f / 362880.0f);
  ps = __fmaf_rn(ps, a2, -1.0f / 5040.0f);
  ps = __fmaf_rn(ps, a2, 1.0f / 120.0f);
  ps = __fmaf_rn(ps, a2, -1.0f / 6.0f);
  ps = __fmaf_rn(ps, a2, 1.0f);
  s = __fmul_rn(ps, a);
  float pc = 1.0f / 20922789888000.0f;
  pc = __fmaf_rn(pc, a2, -1.0f / 87178291200.0f);
  pc = __fmaf_rn(pc, a2, 1.0f / 479001600.0f);
  pc = __fmaf_rn(pc, a2, -1.0f / 3628800.0f);
  pc = __fmaf_rn(pc, a2, 1.0f / 40320.0f);
  pc = __fmaf_rn(pc, a2, -1.0f / 720.0f);
  pc = __fmaf_rn(pc, a2, 1.0f / 24.0f);
  pc = __fmaf_rn(pc, a2, -0.5f);
  pc = __fmaf_rn(pc, a2, 1.0f);
  c = pc;
}

__device__ __forceinline__ void fb_box_muller(uint32_t r0, uint32_t r1, float &z0, float &z1) {
  float u1 = __fmul_rn((float)((r0 >> 8) + 1u), 5.9604644775390625e-08f);
  uint32_t q = r1 >> 30;
  uint32_t fr = (r1 & 0x3FFFFFFFu) >> 6;
  float a = __fmul_rn((float)fr, 9.36227702e-08f);
  float s, c;
  fb_sincos_q(a, s, c);
  float rr = sqrtf(__fmul_rn(-2.0f, fb_ln_u(u1)));  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is the native approx
  float cs, sn;
  if (q == 0) { cs = c; sn = s; }
  else if (q == 1) { cs = -s; sn = c; }
  else if (q == 2) { cs = -c; sn = -s; }
  else { cs = s; sn = -c; }
  z0 = __fmul_rn(rr, cs);
  z1 = __fmul_rn(rr, sn);
}

// 4 consecutive normals z[4*n4 .. 4*n4+3] of antithetic pair j
__device__ __forceinline__ void fb_noise4(uint64_t seed, uint32_t iter, uint32_t stream, uint32_t n4,
                                          uint32_t j, float z[4]) {
  uint32_t r[4];
  fb_philox4x32_10(n4, j, iter, stream, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  fb_box_muller(r[0], r[1], z[0], z[1]);
  fb_box_muller(r[2], r[3], z[2], z[3]);
}

// (x * 2^(bits-1)).astype(int16): trunc toward zero, keep the low 16 bits
// (gmm_ubm_OSI.py:83-85; golden G5: 1.0 -> -32768)
__device__ __forceinline__ int16_t fb_quantize(double x, double scale) {
  double v = __dmul_rn(x, scale);
  long long t;
  if (!(v > -9.2e18 && v < 9.2e18)) t = 0;
  else t = (long long)v;
  return (int16_t)(uint16_t)((unsigned long long)t & 0xFFFFull);
}

// Natural logarithm of a positive, finite, normal double (the front-end only takes logs of energies >= FLT_EPSILON).
// x = m 2^e with m in [sqrt(1/2), sqrt(2)),  log m = 2 atanh(s),  s = (m-1)/(m+1),  |s| <= 0.1716: ten terms of the
// odd series leave a truncation error below 1e-18; e ln2 is added as hi + lo.  Within 2 ulp of a correctly rounded
// log (checked against numpy on 3e6 arguments) in ~45 branch-free instructions (the device
// library's log(): k_mfcc_r4 43.6 us, this one 42.4 us).
__device__ __forceinline__ double fb_log_f64(double x) {
  int e;
  double m = __builtin_frexp(x, &e);  // m in [0.5, 1)
  if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
  const double s = (m - 1.0) / (m + 1.0);
  const double z = s * s;
  double p = 1.0 / 21.0;
  p = __builtin_fma(p, z, 1.0 / 19.0);
  p = __builtin_fma(p, z, 1.0 / 17.0);
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  const double two_s = s + s;
  const double lm = __builtin_fma(two_s * z, p, two_s);
  const double ed = (double)e;
  return __builtin_fma(ed, 0x1.62e42fee00000p-1, __builtin_fma(ed, 0x1.a39ef35793c76p-33, lm));
}

__device__ __forceinline__ double fb_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// A float32 score as the reference sees it after Kaldi's text output (6 significant digits, operator<<)
// and Python's float(): nearest double of the 6-digit decimal.  Exact powers of ten from a table and one
// correctly rounded division, so host (oracle) and device agree bit for bit.
__host__ __device__ inline double fb_round6(double xin) {
  const double x = (double)(float)xin;
  if (x == 0.0 || !(x == x) || x - x != 0.0) return x;
  const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                          1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const double ax = x < 0.0 ? -x : x;
  int e = 0;  // 10^e <= ax < 10^(e+1)
  if (ax >= 1.0) { while (e < 21 && ax >= p10[e + 1]) ++e; }
  else { while (e > -16 && ax * p10[-e] < 1.0) --e; }
  const int k = 5 - e;  // scale to 6 integer digits
  double scaled = k >= 0 ? ax * p10[k > 22 ? 22 : k] : ax / p10[-k];
  double r = rint(scaled);
  int kk = k;
  if (r >= 1e6) { r = rint(r / 10.0); kk -= 1; }
  const double v = kk >= 0 ? r / p10[kk > 22 ? 22 : kk] : r * p10[-kk];
  return x < 0.0 ? -v : v;
}

// DPP row/bank reductions (VALU data path, no LDS round trips): after the six steps lane 63 holds
// the wave total, which is broadcast through a scalar register.  Summation order: pairs, quads,
// half rows, rows (lanes of a 16-lane row), then rows 0+1 / 2+3, then halves -- fixed, so the
// result is deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int fb_dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double fb_dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = fb_dpp_i32<CTRL, ROW_MASK>((int)(b & 0xffffffffll));
  const int hi = fb_dpp_i32<CTRL, ROW_MASK>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int fb_wave_sum_i32_dpp(int v) {
  v += fb_dpp_i32<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += fb_dpp_i32<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += fb_dpp_i32<0x141, 0xf>(v);  // row_half_mirror
  v += fb_dpp_i32<0x140, 0xf>(v);  // row_mirror
  v += fb_dpp_i32<0x142, 0xa>(v);  // row_bcast15 -> rows 1, 3
  v += fb_dpp_i32<0x143, 0xc>(v);  // row_bcast31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double fb_wave_sum_dpp(double v) {
  v += fb_dpp_f64<0xb1, 0xf>(v);
  v += fb_dpp_f64<0x4e, 0xf>(v);
  v += fb_dpp_f64<0x141, 0xf>(v);
  v += fb_dpp_f64<0x140, 0xf>(v);
  v += fb_dpp_f64<0x142, 0xa>(v);
  v += fb_dpp_f64<0x143, 0xc>(v);
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double fb_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}
