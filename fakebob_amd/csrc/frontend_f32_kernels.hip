// frontend_f32_kernels.hip -- k_mfcc_f32: compute-mfcc-feats (gmm_ubm_kaldiHelper.py:138-140) in Kaldi's own
// precision.  Kaldi's BaseFloat front-end is float32 end to end; fb_frontend_cfg.mfcc_f32 = 1
// selects this kernel instead of the float64-between-storage-points k_mfcc_r16 (frontend_kernels.hip).
//
// What stays wider than float32, and why: the frame's raw log-energy C0 -- the only input of the VAD votes
// (compute-vad-decision) and therefore of which frames exist downstream -- is taken from the EXACT energy: the samples
// are integers, so sum x and sum x^2 are exact integers and the energy after DC removal is the single rounding of
// (L sum x^2 - (sum x)^2) / L in float64.  Everything else is float32 with one IEEE rounding per written operation
// (the library is compiled with -ffp-contract=off; fused operations are explicit fmaf): the CPU oracle's twin
// (oracle/fb_oracle.c: fbo_mfcc with cfg.mfcc_f32) performs the same operations in the same order, so the MFCC matrix
// is bit-identical between the two -- summation trees included (the 256-point complex FFT as 16 x 16, radix 4 inside).
//
// Why it is faster than the float64 kernel (30 us per NES batch): not the arithmetic rate -- MI355X issues float64
// vector instructions at the float32 rate -- but occupancy.  Half the registers (a lane's 16 complex points are 32
// registers, not 64) and half the LDS per frame buffer let 16 waves live on a compute unit instead of 8, which is one
// wave per group of four frames for the 15 300 frames of a batch: the whole launch is ONE round (the float64 kernel
// needs two) and four waves per SIMD hide each other's LDS round trips.
//
// Operation order (shared with the oracle twin), lane t of the 16 lanes of a frame, points p = 16 a + t (a = 0 .. 15),
// samples s0 = 32 a + 2 t and s0 + 1:
//   mean  = (float)(sum x) / (float)L                                   [one float32 division]
//   d(s)  = (float)x[s] - mean ;  y[s] = (d(s) - pre * d(s - 1)) * win[s]   (d(-1) := d(0): Kaldi's x[0] -= pre x[0])
//   z[p]  = (y[2p], y[2p + 1])  ->  256-point complex DFT: dft16 over a, times W256^(t k1), dft16 over t
//   cmul(a, w) = (fmaf(-a.y, w.y, a.x * w.x), fmaf(a.y, w.x, a.x * w.y))
//   real-FFT unpack of bin k (0 .. 256) from Z[k], Z[256 - k], power = 0.25 (xr^2 + xi^2)   [formulas in the code]
//   mel_m = sum_i fmaf(w_m[i], pow[first_m + i], .) in order; log as (float)log_f64((double)max(mel, FLT_EPSILON))
//   c_k   = (sum_m fmaf(dct[k][m], logmel[m], .) in order) * lifter[k] ;  c_0 <- (float)log_f64(max(energy, FLT_EPSILON))
#include <float.h>

#include "fb_device.h"
#include "fb_kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FB_F32_WAVES 16
#define FB_F32_SLOTS 272  // complex slots per frame buffer: 256 + one pad per 16 (conflict-free 16 x 16 transpose)

struct MfccF32Lds {  // offsets in floats
  int tw, twf, win, melw, dct, lift, melidx, wave0;
};
__host__ __device__ inline MfccF32Lds fb_mfcc_f32_layout(int L, int nb, int nc, int melw_n) {
  MfccF32Lds o;
  int off = 0;
  o.tw = off; off += 2 * 256;           // exp(-2 pi i m / 256), m < 256
  o.twf = off; off += 2 * 257;          // exp(-2 pi i k / 512), k <= 256
  off = (off + 1) & ~1;
  o.win = off; off += (L + 1) & ~1;
  o.melw = off; off += (melw_n + 2) & ~1;
  o.dct = off; off += (nc * nb + 2) & ~1;
  o.lift = off; off += (nc + 2) & ~1;
  o.melidx = off; off += (3 * nb + 2) & ~1;
  off = (off + 3) & ~3;
  o.wave0 = off;
  return o;
}

__device__ __forceinline__ void fb_wave_sync32() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int fb_f32_phys(int k) { return k + (k >> 4); }
__device__ __forceinline__ int fb_row_sum_i32f(int v) {  // total of the 16 lanes of a DPP row, in every lane
  v += fb_dpp_i32<0xb1, 0xf>(v);
  v += fb_dpp_i32<0x4e, 0xf>(v);
  v += fb_dpp_i32<0x141, 0xf>(v);
  v += fb_dpp_i32<0x140, 0xf>(v);
  return v;
}
__device__ __forceinline__ f32x2 fb_cmul32(f32x2 a, f32x2 w) {
  f32x2 r;
  r.x = __builtin_fmaf(-a.y, w.y, a.x * w.x);
  r.y = __builtin_fmaf(a.y, w.x, a.x * w.y);
  return r;
}
// forward 4-point DFT in place
__device__ __forceinline__ void fb_dft4_32(f32x2 &v0, f32x2 &v1, f32x2 &v2, f32x2 &v3) {
  const f32x2 a = v0 + v2, b = v0 - v2, c = v1 + v3, d = v1 - v3;
  v0 = a + c;
  v2 = a - c;
  v1 = f32x2{b.x + d.y, b.y - d.x};  // b - i d
  v3 = f32x2{b.x - d.y, b.y + d.x};  // b + i d
}
// forward 16-point DFT in registers, natural order in and out: 4 x dft4 over n1 (n = 4 n1 + n2), twiddles W16^(n2 k1),
// 4 x dft4 over n2 (k = k1 + 4 k2)
__device__ __forceinline__ void fb_dft16_32(f32x2 (&v)[16]) {
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) fb_dft4_32(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);  // -> A[n2][k1] at v[4 k1 + n2]
  constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, R2 = 0.70710678118654752440f;
  const f32x2 W1 = {C1, -S1}, W3 = {S1, -C1}, W9 = {-C1, S1};   // W16^m = (cos(pi m / 8), -sin(pi m / 8))
  auto mul_w2 = [&](f32x2 a) { return f32x2{R2 * (a.x + a.y), R2 * (a.y - a.x)}; };      // (1 - i) / sqrt 2
  auto mul_w4 = [&](f32x2 a) { return f32x2{a.y, -a.x}; };                              // -i
  auto mul_w6 = [&](f32x2 a) { return f32x2{R2 * (a.y - a.x), -(R2 * (a.x + a.y))}; };   // -(1 + i) / sqrt 2
  v[4 * 1 + 1] = fb_cmul32(v[4 * 1 + 1], W1);
  v[4 * 1 + 2] = mul_w2(v[4 * 1 + 2]);
  v[4 * 1 + 3] = fb_cmul32(v[4 * 1 + 3], W3);
  v[4 * 2 + 1] = mul_w2(v[4 * 2 + 1]);
  v[4 * 2 + 2] = mul_w4(v[4 * 2 + 2]);
  v[4 * 2 + 3] = mul_w6(v[4 * 2 + 3]);
  v[4 * 3 + 1] = fb_cmul32(v[4 * 3 + 1], W3);
  v[4 * 3 + 2] = mul_w6(v[4 * 3 + 2]);
  v[4 * 3 + 3] = fb_cmul32(v[4 * 3 + 3], W9);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) fb_dft4_32(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);  // X[k1 + 4 k2] at v[4 k1 + k2]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j) { const f32x2 tmp = v[4 * i + j]; v[4 * i + j] = v[4 * j + i]; v[4 * j + i] = tmp; }
}

// NFULL: the points a < NFULL lie inside the frame for every lane (32 a + 31 < L), their validity selects are dropped
// at compile time (12 of 16 for the recipe's L = 400); 0 = no assumption.
template <int NFULL>
__global__ __launch_bounds__(64 * FB_F32_WAVES, 1) void k_mfcc_f32(FbFrontendDev fe, int melw_n,
                                                                   const int16_t *__restrict__ wav,
                                                                   const int4 *__restrict__ frame_rec,
                                                                   int total_frames, float *__restrict__ mfcc) {
  if (fe.stop && *fe.stop) return;
  extern __shared__ __attribute__((aligned(16))) float smem32[];
  constexpr int NT = 64 * FB_F32_WAVES, Nc = 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t_lane = lane & 15, fq = lane >> 4;
  const int L = fe.L, nb = fe.nb, nc = fe.nc;
  const MfccF32Lds lo = fb_mfcc_f32_layout(L, nb, nc, melw_n);
  f32x2 *s_tw = reinterpret_cast<f32x2 *>(smem32 + lo.tw);
  f32x2 *s_twf = reinterpret_cast<f32x2 *>(smem32 + lo.twf);
  float *s_win = smem32 + lo.win;
  float *s_melw = smem32 + lo.melw;
  float *s_dct = smem32 + lo.dct;
  float *s_lift = smem32 + lo.lift;
  int *s_mfirst = reinterpret_cast<int *>(smem32 + lo.melidx), *s_mlen = s_mfirst + nb, *s_moff = s_mlen + nb;
  f32x2 *X = reinterpret_cast<f32x2 *>(smem32 + lo.wave0) + ((size_t)w * 4 + fq) * FB_F32_SLOTS;  // this frame's buffer
  const int n_groups = (total_frames + 3) >> 2;
  const int w_glob = blockIdx.x * FB_F32_WAVES + w, w_step = gridDim.x * FB_F32_WAVES;
  // samples of points p = 16 a + t of frame 4 g + fq: s0 = 32 a + 2 t and s0 + 1, packed into one register per point
  auto load_group = [&](int g, int tl, int (&xq)[16]) {
    const int fl = 4 * g + fq;
    const int4 rec = frame_rec[fl < total_frames ? fl : total_frames - 1];
    const int64_t abs_start = ((int64_t)(unsigned)rec.x) | ((int64_t)rec.y << 32);
    const int start = rec.z, n = rec.w;
    const bool interior = start >= 0 && start + L <= n;
    if (__all(interior)) {
      const int16_t *fr = wav + abs_start;
#pragma unroll
      for (int a = 0; a < 16; ++a) {  // unconditional loads on clamped indices, masked afterwards
        const int s0 = 32 * a + 2 * tl;
        const int lo16 = fr[a < NFULL ? s0 : min(s0, L - 1)], hi16 = fr[a < NFULL ? s0 + 1 : min(s0 + 1, L - 1)];
        xq[a] = (hi16 << 16) | (lo16 & 0xffff);
      }
    } else {
      const int16_t *wv = wav + (abs_start - start);
      int kk[32];  // samples reflected at the utterance edges: the indices first, then all 32 loads in flight
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int s0 = 32 * a + 2 * tl;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int k = start + min(s0 + u, L - 1);
          k = k < 0 ? -k - 1 : (k >= n ? (int)(2u * (unsigned)n - 1u - (unsigned)k) : k);
          while (k < 0 || k >= n) { if (k < 0) k = -k - 1; else k = (int)(2u * (unsigned)n - 1u - (unsigned)k); }
          kk[2 * a + u] = k;
        }
      }
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int lo16 = wv[kk[2 * a]], hi16 = wv[kk[2 * a + 1]];
        xq[a] = (hi16 << 16) | (lo16 & 0xffff);
      }
    }
  };
  int xn[16];
  if (w_glob < n_groups) load_group(w_glob, t_lane, xn);
  // tables: one float32 blob in this kernel's LDS layout, built by fb_set_frontend from the float64 host tables (the
  // window, mel weights, DCT and lifter are float32 values already: Kaldi stores them as BaseFloat) -- a straight copy,
  // one 16-byte load per thread
  {
    const float4 *src = reinterpret_cast<const float4 *>(fe.f32_tab);
    float4 *dst = reinterpret_cast<float4 *>(smem32);
    for (int i = tid; i < lo.wave0 / 4; i += NT) dst[i] = src[i];
  }
  __syncthreads();
  const float pre = (float)fe.preemph;

  for (int g = w_glob; g < n_groups; g += w_step) {
    int t = t_lane;
    asm volatile("" : "+v"(t));  // (keeps the lane-dependent index arithmetic out of registers across trips)
    const int f = 4 * g + fq;
    const bool fvalid = f < total_frames;
    int xp[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) xp[a] = xn[a];
    if (g + w_step < n_groups) load_group(g + w_step, t, xn);
    // ---- exact integer moments of the frame: sum x (|.| < 2^24) and sum x^2 (< 2^39)
    int isum = 0;
    unsigned long long sq = 0ull;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int s0 = 32 * a + 2 * t;
      const int x0 = (a < NFULL || s0 < L) ? (int)(short)xp[a] : 0, x1 = (a < NFULL || s0 + 1 < L) ? (xp[a] >> 16) : 0;
      isum += x0 + x1;
      sq += (unsigned long long)(unsigned)(x0 * x0) + (unsigned long long)(unsigned)(x1 * x1);
    }
    isum = fb_row_sum_i32f(isum);
    const int sq_lo = fb_row_sum_i32f((int)(sq & 0xfffffull)), sq_hi = fb_row_sum_i32f((int)(sq >> 20));  // 16 x 2^20, 16 x 2^15
    const long long sumsq = ((long long)sq_hi << 20) + (long long)sq_lo;
    const long long dc = fe.remove_dc ? (long long)isum : 0ll;
    const double energy = (double)((long long)L * sumsq - dc * dc) / (double)L;  // exact numerator (< 2^53), one rounding
    const float mean = fe.remove_dc ? (float)isum / (float)L : 0.0f;

    f32x2 v[16];
    int prev_rot = 0;  // row-rotated second samples of point a - 1
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int s0 = 32 * a + 2 * t;
      const int xa0 = (int)(short)xp[a], xa1 = xp[a] >> 16;
      const int rot = fb_dpp_i32<0x121, 0xf>(xa1);  // row_ror:1: lane t gets lane (t - 1) & 15
      const int xprev = t == 0 ? (a == 0 ? xa0 : prev_rot) : rot;  // Kaldi: sample 0 is pre-emphasised with itself
      prev_rot = rot;
      const bool inside = a < NFULL;  // compile time
      const f32x2 wq = *reinterpret_cast<const f32x2 *>(&s_win[inside ? s0 : min(s0, (L - 1) & ~1)]);  // L even: the pair exists
      const float w0 = inside || s0 < L ? wq.x : 0.0f, w1 = inside || s0 + 1 < L ? wq.y : 0.0f;
      const float av = (float)xa0 - mean, cv = (float)xa1 - mean, pm = (float)xprev - mean;
      const float y0 = (av - pre * pm) * w0;
      const float y1 = (cv - pre * av) * w1;
      v[a] = f32x2{inside || s0 < L ? y0 : 0.0f, inside || s0 + 1 < L ? y1 : 0.0f};
    }

    // ---- 256-point FFT = radix-16 over a, twiddle W256^(t k1), transpose, radix-16 over b
    fb_dft16_32(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[k1] = fb_cmul32(v[k1], s_tw[(t * k1) & (Nc - 1)]);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) X[fb_f32_phys(16 * k1 + t)] = v[k1];
    fb_wave_sync32();
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = X[fb_f32_phys(16 * t + b)];
    fb_dft16_32(v);  // v[k2] = Z[t + 16 k2]
    fb_wave_sync32();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) X[fb_f32_phys(t + 16 * k2)] = v[k2];
    fb_wave_sync32();
    // ---- real-FFT unpack + power spectrum of bins k = t + 16 k2 (and bin 256 in lane t = 0)
    float pwv[17];
#pragma unroll
    for (int k2 = 0; k2 < 17; ++k2) {
      const int kc = k2 < 16 ? t + 16 * k2 : Nc;
      const f32x2 zk = k2 < 16 ? v[k2 & 15] : X[fb_f32_phys(0)];
      const f32x2 zr = X[fb_f32_phys((Nc - kc) & (Nc - 1))];
      // X[k] = E + W O with E = (Z[k] + conj Z[N-k]) / 2, O = -i (Z[k] - conj Z[N-k]) / 2; the halvings applied once,
      // as 1/4 of the power (exact)
      const float er = zk.x + zr.x, ei = zk.y - zr.y;
      const float dr = zk.x - zr.x, di = zk.y + zr.y;
      const f32x2 wk = s_twf[kc];
      const float xr = er + __builtin_fmaf(wk.x, di, wk.y * dr);
      const float xi = ei + __builtin_fmaf(wk.y, di, -(wk.x * dr));
      pwv[k2] = 0.25f * __builtin_fmaf(xr, xr, xi * xi);
    }
    fb_wave_sync32();
    float *PW = reinterpret_cast<float *>(X);  // 257 floats; LM behind it
    float *LM = PW + 264;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) PW[t + 16 * k2] = pwv[k2];
    if (t == 0) PW[Nc] = pwv[16];
    fb_wave_sync32();
    // ---- mel filterbank + log: lane t takes filters t and t + 16; the free second slot of lane 15 (nb <= 31) takes
    //      the log of the frame energy
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int m = t + 16 * half;
      float e = 0.0f;
      if (m < nb) {
        const float *wm = s_melw + s_moff[m];
        const int first = s_mfirst[m], len = s_mlen[m];
#pragma unroll 4
        for (int i = 0; i < len; ++i) e = __builtin_fmaf(wm[i], PW[first + i], e);
      }
      const bool is_energy = half == 1 && t == 15;
      double ed = is_energy ? energy : (double)e;
      if (ed < (double)FLT_EPSILON) ed = (double)FLT_EPSILON;
      const double le = fb_log_f64(ed);
      if (m < nb) LM[m] = (float)le;
      if (is_energy) LM[nb] = (float)(le < fe.log_energy_floor ? fe.log_energy_floor : le);
    }
    fb_wave_sync32();
    // ---- DCT-II, lifter, C0 <- log energy: lane t takes coefficients t and t + 16
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int c = t + 16 * half;
      if (c < nc) {
        const float *dr = s_dct + c * nb;
        float acc = 0.0f;
#pragma unroll 6
        for (int m = 0; m < nb; ++m) acc = __builtin_fmaf(dr[m], LM[m], acc);
        float o = acc * s_lift[c];
        if (c == 0 && fe.use_energy) o = LM[nb];
        if (fvalid) mfcc[(size_t)f * nc + c] = o;
      }
    }
    fb_wave_sync32();
  }
}

std::vector<float> fb_mfcc_f32_table(int L, int nb, int nc, const double *window, const double *tw_half, const double *tw_full,
                                     const int *mel_first, const int *mel_len, const int *mel_off, const double *mel_w, int melw_n,
                                     const double *dct, const double *lifter) {
  const MfccF32Lds lo = fb_mfcc_f32_layout(L, nb, nc, melw_n);
  std::vector<float> t((size_t)lo.wave0 + 4, 0.0f);
  for (int i = 0; i < 256; ++i) { t[lo.tw + 2 * i] = (float)tw_half[2 * i]; t[lo.tw + 2 * i + 1] = (float)tw_half[2 * i + 1]; }
  for (int i = 0; i <= 256; ++i) { t[lo.twf + 2 * i] = (float)tw_full[2 * i]; t[lo.twf + 2 * i + 1] = (float)tw_full[2 * i + 1]; }
  for (int i = 0; i < L; ++i) t[lo.win + i] = (float)window[i];
  for (int i = 0; i < melw_n; ++i) t[lo.melw + i] = (float)mel_w[i];
  for (int i = 0; i < nc * nb; ++i) t[lo.dct + i] = (float)dct[i];
  for (int i = 0; i < nc; ++i) t[lo.lift + i] = (float)lifter[i];
  int *mi = reinterpret_cast<int *>(&t[lo.melidx]);
  for (int i = 0; i < nb; ++i) { mi[i] = mel_first[i]; mi[nb + i] = mel_len[i]; mi[2 * nb + i] = mel_off[i]; }
  return t;
}
// true when the configuration is one k_mfcc_f32 takes (the recipe's: P = 512, raw energy, at most 31 mel bins / 32
// cepstra, even frame length); fb_set_frontend refuses mfcc_f32 = 1 otherwise
bool fb_mfcc_f32_supported(const FbFrontendDev &fe) {
  return fe.P == 512 && fe.nb <= 31 && fe.nc <= 32 && (fe.L & 1) == 0 && fe.L >= 2 && fe.L <= 512 && fe.raw_energy != 0 && fe.f32_tab != nullptr;
}
bool fb_launch_mfcc_f32(hipStream_t s, const FbFrontendDev &fe, int melw_n, const int16_t *wav, const int32_t *frame_rec,
                        int total_frames, float *mfcc) {
  if (total_frames <= 0) return true;
  if (!fb_mfcc_f32_supported(fe)) return false;
  const MfccF32Lds l = fb_mfcc_f32_layout(fe.L, fe.nb, fe.nc, melw_n);
  const size_t shm = sizeof(float) * (size_t)l.wave0 + sizeof(f32x2) * (size_t)FB_F32_WAVES * 4 * FB_F32_SLOTS;
  if (shm > 160 * 1024) return false;
  static std::atomic<unsigned long long> optin{0};
  unsigned long long bit = 0;
  if (fb_device_needs_optin(optin, &bit)) {
    const void *fns[] = {reinterpret_cast<const void *>(k_mfcc_f32<12>), reinterpret_cast<const void *>(k_mfcc_f32<0>)};
    for (const void *fn : fns)
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    optin.fetch_or(bit, std::memory_order_release);
  }
  const int n_groups = (total_frames + 3) / 4;
  const int rounds = (n_groups + 256 * FB_F32_WAVES - 1) / (256 * FB_F32_WAVES);
  const int blocks = (n_groups + rounds * FB_F32_WAVES - 1) / (rounds * FB_F32_WAVES);
  const int4 *rec = reinterpret_cast<const int4 *>(frame_rec);
  if (fe.L / 32 >= 12) hipLaunchKernelGGL((k_mfcc_f32<12>), dim3(blocks), dim3(64 * FB_F32_WAVES), shm, s, fe, melw_n, wav, rec, total_frames, mfcc);
  else hipLaunchKernelGGL((k_mfcc_f32<0>), dim3(blocks), dim3(64 * FB_F32_WAVES), shm, s, fe, melw_n, wav, rec, total_frames, mfcc);
  return true;
}
