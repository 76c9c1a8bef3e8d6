// frontend_kernels.hip -- Kaldi-equivalent speech front-end on gfx950:
//   K2 compute-mfcc-feats (framing w/ reflection, DC removal, raw log-energy,
//      pre-emphasis, Povey window, real FFT via a half-size complex Stockham
//      FFT in LDS, mel filterbank, log, DCT-II, lifter, C0 <- log-energy)
//   K3 compute-vad-decision      K4 add-deltas
//   K5 apply-cmvn-sliding (center, mean only)  K6 select-voiced-frames
// The reference runs these as external programs (gmm_ubm_kaldiHelper.py:131-169,
// :195-198); the algorithms are restated from SURVEY.md Appendix A ([EXT]).
// Precision policy = the oracle's: float64 between Kaldi's float32 storage points.
#include <float.h>

#include <algorithm>
#include <cstring>
#include <cstdlib>

#include "fb_device.h"
#include "fb_kernels.h"

__device__ __forceinline__ int fb_find_utt(const int *__restrict__ frame_off, int B, int f) {
  int lo = 0, hi = B;  // frame_off[lo] <= f < frame_off[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (frame_off[mid] <= f) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void fb_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------- MFCC
// One wave per frame, 4 frames in flight per 256-thread workgroup, workgroups persistent over
// frames.  LDS: the constant tables (window, FFT twiddles, packed mel weights, DCT, lifter) are
// staged once per workgroup; each wave owns two P-double ping-pong buffers (complex FFT of size
// P/2 + real-FFT unpack; power spectrum and log-mel energies alias them).
#define FB_MFCC_MAXI 8  // samples per lane: frame_length <= 512

struct MfccLds {  // offsets in doubles from the dynamic LDS base
  int tw, win, melw, dct, lift, melidx, wave0, per_wave;
};
__host__ __device__ inline MfccLds fb_mfcc_layout(int P, int L, int nb, int nc, int melw_n) {
  MfccLds o;
  int off = 0;
  o.tw = off; off += P;                        // (P/2) double2
  o.win = off; off += (L + 1) / 2;             // L floats
  o.melw = off; off += (melw_n + 1) / 2 + 1;   // packed mel weights, floats
  o.dct = off; off += (nc * nb + 1) / 2 + 1;   // floats
  o.lift = off; off += (nc + 1) / 2 + 1;       // floats
  o.melidx = off; off += (3 * nb + 1) / 2 + 1; // ints: first, len, off
  off = (off + 1) & ~1;
  o.wave0 = off;
  o.per_wave = 2 * P;
  return o;
}

__global__ __launch_bounds__(256) void k_mfcc(FbFrontendDev fe, int melw_n, const int16_t *__restrict__ wav,
                                              const int64_t *__restrict__ wav_off,
                                              const int *__restrict__ frame_off, int B, int total_frames,
                                              float *__restrict__ mfcc) {
  if (fe.stop && *fe.stop) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int P = fe.P, Nc = P >> 1, L = fe.L, nb = fe.nb, nc = fe.nc;
  const MfccLds lo = fb_mfcc_layout(P, L, nb, nc, melw_n);
  double2 *s_tw = reinterpret_cast<double2 *>(smem + lo.tw);
  float *s_win = reinterpret_cast<float *>(smem + lo.win);
  float *s_melw = reinterpret_cast<float *>(smem + lo.melw);
  float *s_dct = reinterpret_cast<float *>(smem + lo.dct);
  float *s_lift = reinterpret_cast<float *>(smem + lo.lift);
  int *s_mfirst = reinterpret_cast<int *>(smem + lo.melidx), *s_mlen = s_mfirst + nb, *s_moff = s_mlen + nb;
  double *A = smem + lo.wave0 + (size_t)w * lo.per_wave;
  double *Bf = A + P;
  // ---- stage the tables (float32-valued constants are stored as float in LDS)
  for (int i = tid; i < Nc; i += 256) s_tw[i] = reinterpret_cast<const double2 *>(fe.tw_half)[i];
  for (int i = tid; i < L; i += 256) s_win[i] = (float)fe.window[i];
  for (int i = tid; i < melw_n; i += 256) s_melw[i] = (float)fe.mel_w[i];
  for (int i = tid; i < nc * nb; i += 256) s_dct[i] = (float)fe.dct[i];
  for (int i = tid; i < nc; i += 256) s_lift[i] = (float)fe.lifter[i];
  for (int i = tid; i < nb; i += 256) { s_mfirst[i] = fe.mel_first[i]; s_mlen[i] = fe.mel_len[i]; s_moff[i] = fe.mel_off[i]; }
  __syncthreads();
  const double2 *twf = reinterpret_cast<const double2 *>(fe.tw_full);

  // Waves are independent from here on: each owns its LDS buffers and walks its own frames, so
  // only wave-level ordering is needed (LDS executes one wave's DS ops in order; the fence +
  // wave_barrier pair stops the compiler from moving LDS accesses across the phase boundary).
  for (int f = blockIdx.x * 4 + w; f < total_frames; f += gridDim.x * 4) {
    const bool valid = true;
    const int b = fb_find_utt(frame_off, B, f);
    const int t = f - frame_off[b];
    const int64_t n = wav_off[b + 1] - wav_off[b];
    const int16_t *wv = wav + wav_off[b];
    const int64_t start = fe.snip_edges ? (int64_t)t * fe.shift : (int64_t)t * fe.shift + fe.shift / 2 - L / 2;

    // ---- load (reflect at the edges), DC removal, raw energy
    double xs[FB_MFCC_MAXI];
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < FB_MFCC_MAXI; ++i) {
      const int s = lane + 64 * i;
      double v = 0.0;
      if (s < L) {
        int64_t k = start + s;
        while (k < 0 || k >= n) { if (k < 0) k = -k - 1; else k = 2 * n - 1 - k; }
        v = (double)wv[k];
      }
      xs[i] = v;
      sum += v;  // integers: exact in any order
    }
    sum = fb_wave_sum(sum);
    const double mean = fe.remove_dc ? sum / (double)L : 0.0;
    double en = 0.0;
#pragma unroll
    for (int i = 0; i < FB_MFCC_MAXI; ++i) {
      const int s = lane + 64 * i;
      if (s < L) { xs[i] -= mean; en = fma(xs[i], xs[i], en); Bf[s] = xs[i]; }
    }
    double energy = fb_wave_sum(en);
    fb_wave_sync();
    // ---- pre-emphasis + window, packed as P/2 complex points (re=y[2k], im=y[2k+1])
    double en2 = 0.0;
#pragma unroll
    for (int i = 0; i < FB_MFCC_MAXI; ++i) {
      const int s = lane + 64 * i;
      if (s < P) {
        double y = 0.0;
        if (s < L) {
          const double prev = Bf[s > 0 ? s - 1 : 0];
          y = (xs[i] - fe.preemph * prev) * (double)s_win[s];
          en2 = fma(y, y, en2);
        }
        A[s] = y;
      }
    }
    if (!fe.raw_energy) energy = fb_wave_sum(en2);
    double log_energy = fb_log_f64(energy > (double)FLT_EPSILON ? energy : (double)FLT_EPSILON);
    if (log_energy < fe.log_energy_floor) log_energy = fe.log_energy_floor;
    fb_wave_sync();
    // ---- Stockham radix-2 complex FFT of size Nc (ping-pong A <-> Bf)
    double2 *src = reinterpret_cast<double2 *>(A), *dst = reinterpret_cast<double2 *>(Bf);
    for (int Ns = 1; Ns < Nc; Ns <<= 1) {
      const int tstep = Nc / (2 * Ns);
      for (int j = lane; j < Nc / 2; j += 64) {
        const int k = j & (Ns - 1);
        const double2 wv2 = s_tw[k * tstep];
        const double2 v0 = src[j], x1 = src[j + Nc / 2];
        double2 v1;
        v1.x = x1.x * wv2.x - x1.y * wv2.y;
        v1.y = x1.x * wv2.y + x1.y * wv2.x;
        const int idx = ((j - k) << 1) + k;
        dst[idx] = make_double2(v0.x + v1.x, v0.y + v1.y);
        dst[idx + Ns] = make_double2(v0.x - v1.x, v0.y - v1.y);
      }
      fb_wave_sync();
      double2 *tmp = src; src = dst; dst = tmp;
    }
    // ---- real-FFT unpack + power spectrum, bins 0..Nc  -> PW (aliases dst)
    double pwv[FB_MFCC_MAXI];
#pragma unroll
    for (int i = 0; i < FB_MFCC_MAXI; ++i) {
      const int k = lane + 64 * i;
      pwv[i] = 0.0;
      if (k <= Nc) {
        const double2 zk = src[k & (Nc - 1)];
        const double2 zr = src[(Nc - k) & (Nc - 1)];
        const double er = 0.5 * (zk.x + zr.x), ei = 0.5 * (zk.y - zr.y);  // E = (Zk + conj(Zr))/2
        const double dr = zk.x - zr.x, di = zk.y + zr.y;                  // d = Zk - conj(Zr)
        const double orr = 0.5 * di, oi = -0.5 * dr;                      // O = -i/2 * d
        const double2 wk = twf[k];
        const double xr = er + (wk.x * orr - wk.y * oi);
        const double xi = ei + (wk.x * oi + wk.y * orr);
        pwv[i] = xr * xr + xi * xi;
      }
    }
    double *PW = reinterpret_cast<double *>(dst);
#pragma unroll
    for (int i = 0; i < FB_MFCC_MAXI; ++i) {
      const int k = lane + 64 * i;
      if (k <= Nc) PW[k] = pwv[i];
    }
    fb_wave_sync();
    // ---- mel filterbank + log: two lanes per filter, LM aliases src
    double *LM = reinterpret_cast<double *>(src);
    for (int m2 = lane; m2 < 2 * ((nb + 31) / 32) * 32; m2 += 64) {
      const int m = m2 >> 1, part = m2 & 1;
      double e = 0.0;
      if (m < nb) {
        const float *wm = s_melw + s_moff[m];
        const int first = s_mfirst[m], len = s_mlen[m];
        const int h0 = (len + 1) >> 1;
        const int i0 = part ? h0 : 0, i1 = part ? len : h0;
        for (int i = i0; i < i1; ++i) e = fma((double)wm[i], PW[first + i], e);
      }
      e += __shfl_xor(e, 1, 64);
      if (m < nb && part == 0) {
        if (e < (double)FLT_EPSILON) e = (double)FLT_EPSILON;
        LM[m] = fb_log_f64(e);
      }
    }
    fb_wave_sync();
    // ---- DCT-II, lifter, C0 <- log energy: two lanes per coefficient
    for (int c2 = lane; c2 < 2 * ((nc + 31) / 32) * 32; c2 += 64) {
      const int c = c2 >> 1, part = c2 & 1;
      double acc = 0.0;
      if (c < nc) {
        const float *dr = s_dct + c * nb;
        const int h0 = (nb + 1) >> 1;
        const int i0 = part ? h0 : 0, i1 = part ? nb : h0;
        for (int m = i0; m < i1; ++m) acc = fma((double)dr[m], LM[m], acc);
      }
      acc += __shfl_xor(acc, 1, 64);
      if (c < nc && part == 0) {
        acc *= (double)s_lift[c];
        float o = (float)acc;
        if (c == 0 && fe.use_energy) o = (float)log_energy;
        if (valid) mfcc[(size_t)f * nc + c] = o;
      }
    }
    fb_wave_sync();  // buffers are reused by the next frame
  }
}

// ---------------------------------------------------------- MFCC, P = 512 (the recipe's size)
// LDS table layout and small FFT helpers of k_mfcc_r16 (P = 512).
#define FB_R4_XSLOTS 288  // 256 + 256/8 padded complex slots (per-wave scratch size the layout reserves)

struct MfccR4Lds {  // offsets in doubles
  int tw, twf, win, melw, dct, lift, melidx, wave0, per_wave;
};
__host__ __device__ inline MfccR4Lds fb_mfcc_r4_layout(int L, int nb, int nc, int melw_n) {
  MfccR4Lds o;
  int off = 0;
  o.tw = off; off += 2 * 256;                  // exp(-2 pi i m / 256), m < 256
  o.twf = off; off += 2 * 257;                 // exp(-2 pi i k / 512), k <= 256
  o.win = off; off += (L + 1) / 2;             // floats
  o.melw = off; off += (melw_n + 1) / 2 + 1;
  o.dct = off; off += (nc * nb + 1) / 2 + 1;
  o.lift = off; off += (nc + 1) / 2 + 1;
  o.melidx = off; off += (3 * nb + 1) / 2 + 1;
  off = (off + 1) & ~1;
  o.wave0 = off;
  o.per_wave = 2 * FB_R4_XSLOTS;
  return o;
}

__device__ __forceinline__ double2 fb_cmul(double2 a, double2 w) {
  return make_double2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}
// forward 4-point DFT in place
__device__ __forceinline__ void fb_dft4(double2 &v0, double2 &v1, double2 &v2, double2 &v3) {
  const double2 a = make_double2(v0.x + v2.x, v0.y + v2.y), b = make_double2(v0.x - v2.x, v0.y - v2.y);
  const double2 c = make_double2(v1.x + v3.x, v1.y + v3.y), d = make_double2(v1.x - v3.x, v1.y - v3.y);
  v0 = make_double2(a.x + c.x, a.y + c.y);
  v2 = make_double2(a.x - c.x, a.y - c.y);
  v1 = make_double2(b.x + d.y, b.y - d.x);  // b - i d
  v3 = make_double2(b.x - d.y, b.y + d.x);  // b + i d
}

// ------------------------------------------------------------------------------------------------
// k_mfcc_r16: P = 512 with FOUR frames per wave.  16 lanes own one frame and 16 complex points each, so the 256-point
// complex FFT is two radix-16 passes held in registers with ONE transpose through LDS between them (a radix-4 form needs three exchanges
// per frame), and every LDS / DPP round trip of the remaining stages serves four frames.  Same arithmetic
// (float64 between Kaldi's float32 storage points), different summation trees than the generic k_mfcc -- both sit within 1e-6 of
// the oracle.  16 lanes = one DPP row: the per-frame reductions are four row-local DPP steps.
#define FB_R16_WAVES 8
#define FB_R16_SLOTS 272  // complex slots per frame buffer: 256 + one pad per 16 (conflict-free 16 x 16 transpose)
__device__ __forceinline__ int fb_r16_phys(int k) { return k + (k >> 4); }
__device__ __forceinline__ int fb_row_sum_i32(int v) {  // total of the 16 lanes of a DPP row, in every lane
  v += fb_dpp_i32<0xb1, 0xf>(v);
  v += fb_dpp_i32<0x4e, 0xf>(v);
  v += fb_dpp_i32<0x141, 0xf>(v);
  v += fb_dpp_i32<0x140, 0xf>(v);
  return v;
}
__device__ __forceinline__ double fb_row_sum_f64(double v) {
  v += fb_dpp_f64<0xb1, 0xf>(v);
  v += fb_dpp_f64<0x4e, 0xf>(v);
  v += fb_dpp_f64<0x141, 0xf>(v);
  v += fb_dpp_f64<0x140, 0xf>(v);
  return v;
}
// forward 16-point DFT in registers, natural order in and out: 4 x dft4 over n1 (n = 4 n1 + n2), twiddles
// W16^(n2 k1), 4 x dft4 over n2 (k = k1 + 4 k2)
__device__ __forceinline__ void fb_dft16(double2 (&v)[16]) {
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) fb_dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);  // -> A[n2][k1] at v[4 k1 + n2]
  constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, R2 = 0.70710678118654752440;
  // W16^m = (cos(pi m / 8), -sin(pi m / 8)), m = n2 * k1
  const double2 W1 = make_double2(C1, -S1), W3 = make_double2(S1, -C1), W9 = make_double2(-C1, S1);
  // W2 = (1 - i) / sqrt 2, W4 = -i, W6 = -(1 + i) / sqrt 2: a swap, or two additions and two multiplications, instead of
  // the general product's four and two (the compiler may not fold a multiplication by 0 or by equal constants)
  auto mul_w2 = [&](double2 a) { return make_double2(R2 * (a.x + a.y), R2 * (a.y - a.x)); };
  auto mul_w4 = [&](double2 a) { return make_double2(a.y, -a.x); };
  auto mul_w6 = [&](double2 a) { return make_double2(R2 * (a.y - a.x), -(R2 * (a.x + a.y))); };
  v[4 * 1 + 1] = fb_cmul(v[4 * 1 + 1], W1);
  v[4 * 1 + 2] = mul_w2(v[4 * 1 + 2]);
  v[4 * 1 + 3] = fb_cmul(v[4 * 1 + 3], W3);
  v[4 * 2 + 1] = mul_w2(v[4 * 2 + 1]);
  v[4 * 2 + 2] = mul_w4(v[4 * 2 + 2]);
  v[4 * 2 + 3] = mul_w6(v[4 * 2 + 3]);
  v[4 * 3 + 1] = fb_cmul(v[4 * 3 + 1], W3);
  v[4 * 3 + 2] = mul_w6(v[4 * 3 + 2]);
  v[4 * 3 + 3] = fb_cmul(v[4 * 3 + 3], W9);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) fb_dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);  // X[k1 + 4 k2] at v[4 k1 + k2]
  // natural order: out[k1 + 4 k2] <- v[4 k1 + k2]  (a 4 x 4 transpose of register names)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j) { const double2 tmp = v[4 * i + j]; v[4 * i + j] = v[4 * j + i]; v[4 * j + i] = tmp; }
}

// NFULL: the points a < NFULL lie inside the frame for every lane (32 a + 31 < L: NFULL <= L / 32), so their window /
// validity selects and mask multiplications are dropped at compile time (12 of 16 points for the recipe's L = 400:
// ~ 12 % of a group's instructions; the kernel is bound by its float64 instruction count); 0 = no assumption.
// RAW: which of the two frame energies is wanted (--raw-energy): the other one's 32 fmas are not issued.
template <int NFULL, bool RAW>
__global__ __launch_bounds__(64 * FB_R16_WAVES, 1) void k_mfcc_r16(FbFrontendDev fe, int melw_n,
                                                                   const int16_t *__restrict__ wav,
                                                                   const int4 *__restrict__ frame_rec,
                                                                   int total_frames, float *__restrict__ mfcc) {
  if (fe.stop && *fe.stop) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NT = 64 * FB_R16_WAVES, Nc = 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = lane & 15, fq = lane >> 4;
  const int L = fe.L, nb = fe.nb, nc = fe.nc;
  const MfccR4Lds lo = fb_mfcc_r4_layout(L, nb, nc, melw_n);  
  double2 *s_tw = reinterpret_cast<double2 *>(smem + lo.tw);
  double2 *s_twf = reinterpret_cast<double2 *>(smem + lo.twf);
  float *s_win = reinterpret_cast<float *>(smem + lo.win);
  float *s_melw = reinterpret_cast<float *>(smem + lo.melw);
  float *s_dct = reinterpret_cast<float *>(smem + lo.dct);
  float *s_lift = reinterpret_cast<float *>(smem + lo.lift);
  int *s_mfirst = reinterpret_cast<int *>(smem + lo.melidx), *s_mlen = s_mfirst + nb, *s_moff = s_mlen + nb;
  double2 *X = reinterpret_cast<double2 *>(smem + lo.wave0) + ((size_t)w * 4 + fq) * FB_R16_SLOTS;  // this frame's buffer
  const int n_groups = (total_frames + 3) >> 2;
  const int w_glob = blockIdx.x * FB_R16_WAVES + w, w_step = gridDim.x * FB_R16_WAVES;
  const int t_lane = t;
  // samples of points p = 16 a + t of frame 4 g + fq: s0 = 32 a + 2 t and s0 + 1, packed into one register per point
  // (the pre-emphasis neighbour s0 - 1 is the previous lane's second sample).  The next group's samples are requested
  // before the current group is processed, so their L2 latency is off the critical path.
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
      // samples reflected at the utterance edges
      // (a wave with an edge frame used to walk these 16 points one global round trip at a time -- 16 x ~1.5 us, the
      //  longest path of the whole launch; the indices first, then all 32 loads in flight)
      int kk[32];
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int s0 = 32 * a + 2 * tl;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int k = start + min(s0 + u, L - 1);
          // (2 n - 1 - k in unsigned arithmetic: n < 2^31, the result is in [-(L), n))
          k = k < 0 ? -k - 1 : (k >= n ? (int)(2u * (unsigned)n - 1u - (unsigned)k) : k);  // one reflection: all but tiny utterances
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
  // the first group's samples are requested BEFORE the tables are staged: their two dependent global round trips
  // (frame record, then samples) run in the shadow of the staging loads and the barrier
  int xn[16];
  if (w_glob < n_groups) load_group(w_glob, t_lane, xn);
  for (int i = tid; i < Nc; i += NT) s_tw[i] = reinterpret_cast<const double2 *>(fe.tw_half)[i];
  for (int i = tid; i <= Nc; i += NT) s_twf[i] = reinterpret_cast<const double2 *>(fe.tw_full)[i];
  for (int i = tid; i < L; i += NT) s_win[i] = (float)fe.window[i];
  for (int i = tid; i < melw_n; i += NT) s_melw[i] = (float)fe.mel_w[i];
  for (int i = tid; i < nc * nb; i += NT) s_dct[i] = (float)fe.dct[i];
  for (int i = tid; i < nc; i += NT) s_lift[i] = (float)fe.lifter[i];
  for (int i = tid; i < nb; i += NT) { s_mfirst[i] = fe.mel_first[i]; s_mlen[i] = fe.mel_len[i]; s_moff[i] = fe.mel_off[i]; }
  __syncthreads();

  for (int g = w_glob; g < n_groups; g += w_step) {
    // Everything below that depends only on the lane (clamped sample indices, window weights, twiddles: > 150
    // registers) is loop-invariant; hoisted it spills, so the lane index is laundered through an empty asm per trip.
    int t = t_lane;
    asm volatile("" : "+v"(t));
    const int f = 4 * g + fq;
    const bool fvalid = f < total_frames;
    int xp[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) xp[a] = xn[a];
    if (g + w_step < n_groups) load_group(g + w_step, t, xn);
    int isum = 0;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int s0 = 32 * a + 2 * t;
      isum += (a < NFULL || s0 < L ? (int)(short)xp[a] : 0) + (a < NFULL || s0 + 1 < L ? (xp[a] >> 16) : 0);
    }
    // DC: the samples are integers, |sum| < 2^24: exact in int32 in any order
    const double mean = fe.remove_dc ? (double)fb_row_sum_i32(isum) / (double)L : 0.0;
    double en = 0.0, en2 = 0.0;
    double2 v[16];
    int prev_rot = 0;  // row-rotated second samples of point a - 1
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int s0 = 32 * a + 2 * t;
      const int xa0 = (int)(short)xp[a], xa1 = xp[a] >> 16;
      const int rot = fb_dpp_i32<0x121, 0xf>(xa1);  // row_ror:1: lane t gets lane (t - 1) & 15
      const int xprev = t == 0 ? (a == 0 ? xa0 : prev_rot) : rot;  // Kaldi: sample 0 is pre-emphasised with itself
      prev_rot = rot;
      const bool inside = a < NFULL;  // compile time
      const float2 wq = *reinterpret_cast<const float2 *>(&s_win[inside ? s0 : min(s0, (L - 1) & ~1)]);  // L even: the pair exists
      const double w0 = inside || s0 < L ? (double)wq.x : 0.0, w1 = inside || s0 + 1 < L ? (double)wq.y : 0.0;
      const double m0 = inside || s0 < L ? 1.0 : 0.0, m1 = inside || s0 + 1 < L ? 1.0 : 0.0;
      const double av = inside ? (double)xa0 - mean : ((double)xa0 - mean) * m0;
      const double cv = inside ? (double)xa1 - mean : ((double)xa1 - mean) * m1;
      const double pm = (double)xprev - mean;
      if constexpr (RAW) {
        en = fma(av, av, en);
        en = fma(cv, cv, en);
      }
      const double y0 = (av - fe.preemph * pm) * w0;
      const double y1 = (cv - fe.preemph * av) * w1;
      if constexpr (!RAW) {
        en2 = fma(y0, y0, en2);
        en2 = fma(y1, y1, en2);
      }
      v[a] = make_double2(y0, y1);
    }
    const double energy = fb_row_sum_f64(RAW ? en : en2);  // its log is taken with the mel logs below

    // ---- 256-point FFT = radix-16 over a, twiddle W256^(t k1), transpose, radix-16 over b
    fb_dft16(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
      v[k1] = fb_cmul(v[k1], s_tw[(t * k1) & (Nc - 1)]);
      if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most 4 twiddle loads in flight: bounds the live registers
    }
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) X[fb_r16_phys(16 * k1 + t)] = v[k1];
    fb_wave_sync();
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = X[fb_r16_phys(16 * t + b)];
    fb_dft16(v);  // v[k2] = Z[t + 16 k2]
    fb_wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) X[fb_r16_phys(t + 16 * k2)] = v[k2];
    fb_wave_sync();
    // ---- real-FFT unpack + power spectrum of bins k = t + 16 k2 (and bin 256 in lane t = 0)
    double pwv[17];
#pragma unroll
    for (int k2 = 0; k2 < 17; ++k2) {
      const int kc = k2 < 16 ? t + 16 * k2 : Nc;
      const double2 zk = k2 < 16 ? v[k2 & 15] : X[fb_r16_phys(0)];
      const double2 zr = X[fb_r16_phys((Nc - kc) & (Nc - 1))];
      // X[k] = E + W O with E = (Z[k] + conj Z[N-k]) / 2, O = -i (Z[k] - conj Z[N-k]) / 2: the four halvings are
      // taken out of the sums and applied once, as 1/4 of the power -- exact (a power of two commutes with every
      // rounding here), so the result is the same double
      const double er = zk.x + zr.x, ei = zk.y - zr.y;
      const double dr = zk.x - zr.x, di = zk.y + zr.y;
      const double2 wk = s_twf[kc];
      const double xr = er + (wk.x * di + wk.y * dr);
      const double xi = ei + (wk.y * di - wk.x * dr);
      pwv[k2] = 0.25 * (xr * xr + xi * xi);
      if ((k2 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    fb_wave_sync();
    double *PW = reinterpret_cast<double *>(X);  // 257 doubles; LM behind it
    double *LM = PW + 264;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) PW[t + 16 * k2] = pwv[k2];
    if (t == 0) PW[Nc] = pwv[16];
    fb_wave_sync();
    // ---- mel filterbank + log: lane t takes filters t and t + 16; the free second slot of lane 15 (nb <= 31)
    //      takes the log of the frame energy
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int m = t + 16 * half;
      double e = 0.0;
      if (m < nb) {
        const float *wm = s_melw + s_moff[m];
        const int first = s_mfirst[m], len = s_mlen[m];
#pragma unroll 4
        for (int i = 0; i < len; ++i) e = fma((double)wm[i], PW[first + i], e);
      }
      const bool is_energy = half == 1 && t == 15;
      if (is_energy) e = energy;
      if (e < (double)FLT_EPSILON) e = (double)FLT_EPSILON;
      const double le = fb_log_f64(e);
      if (m < nb) LM[m] = le;
      if (is_energy) LM[nb] = le < fe.log_energy_floor ? fe.log_energy_floor : le;
    }
    fb_wave_sync();
    // ---- DCT-II, lifter, C0 <- log energy: lane t takes coefficients t and t + 16
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int c = t + 16 * half;
      if (c < nc) {
        const float *dr = s_dct + c * nb;
        double acc = 0.0;
#pragma unroll 4
        for (int m = 0; m < nb; ++m) acc = fma((double)dr[m], LM[m], acc);
        acc *= (double)s_lift[c];
        float o = (float)acc;
        if (c == 0 && fe.use_energy) o = (float)LM[nb];
        if (fvalid) mfcc[(size_t)f * nc + c] = o;
      }
    }
    fb_wave_sync();
  }
}

int fb_mfcc_layout_doubles(int P, int L, int nb, int nc, int melw_n) {
  const MfccLds lo = fb_mfcc_layout(P, L, nb, nc, melw_n);
  return lo.wave0 + 4 * lo.per_wave;
}

void fb_launch_mfcc(hipStream_t s, const FbFrontendDev &fe, int melw_n, const int16_t *wav,
                    const int64_t *wav_off, const int *frame_off, const int32_t *frame_rec, int B,
                    int total_frames, float *mfcc) {
  if (total_frames <= 0) return;
  if (fe.P == 512 && fe.nb <= 31 && fe.nc <= 32 && (fe.L & 1) == 0 && fe.L >= 2) {
    const MfccR4Lds l16 = fb_mfcc_r4_layout(fe.L, fe.nb, fe.nc, melw_n);
    const size_t shm16 = sizeof(double) * (size_t)l16.wave0 + sizeof(double2) * (size_t)FB_R16_WAVES * 4 * FB_R16_SLOTS;
    static std::atomic<unsigned long long> optin16{0};
    unsigned long long bit16 = 0;
    bool ok16 = true;
    if (fb_device_needs_optin(optin16, &bit16)) {
      const void *fns[] = {reinterpret_cast<const void *>(k_mfcc_r16<12, true>), reinterpret_cast<const void *>(k_mfcc_r16<12, false>),
                           reinterpret_cast<const void *>(k_mfcc_r16<0, true>), reinterpret_cast<const void *>(k_mfcc_r16<0, false>)};
      for (const void *fn : fns)
        ok16 = ok16 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      if (ok16) optin16.fetch_or(bit16, std::memory_order_release);
    }
    if (ok16 && shm16 <= 160 * 1024) {
      const int n_groups = (total_frames + 3) / 4;
      const int rounds = (n_groups + 256 * FB_R16_WAVES - 1) / (256 * FB_R16_WAVES);
      const int blocks = (n_groups + rounds * FB_R16_WAVES - 1) / (rounds * FB_R16_WAVES);
      const dim3 grid(blocks), blk(64 * FB_R16_WAVES);
      const int4 *rec = reinterpret_cast<const int4 *>(frame_rec);
      const bool full12 = fe.L / 32 >= 12;
      if (full12 && fe.raw_energy) hipLaunchKernelGGL((k_mfcc_r16<12, true>), grid, blk, shm16, s, fe, melw_n, wav, rec, total_frames, mfcc);
      else if (full12) hipLaunchKernelGGL((k_mfcc_r16<12, false>), grid, blk, shm16, s, fe, melw_n, wav, rec, total_frames, mfcc);
      else if (fe.raw_energy) hipLaunchKernelGGL((k_mfcc_r16<0, true>), grid, blk, shm16, s, fe, melw_n, wav, rec, total_frames, mfcc);
      else hipLaunchKernelGGL((k_mfcc_r16<0, false>), grid, blk, shm16, s, fe, melw_n, wav, rec, total_frames, mfcc);
      return;
    }
  }
  const MfccLds lo = fb_mfcc_layout(fe.P, fe.L, fe.nb, fe.nc, melw_n);
  size_t shm = sizeof(double) * (size_t)(lo.wave0 + 4 * lo.per_wave);
  int blocks = (total_frames + 3) / 4;
  if (blocks > 768) blocks = 768;
  hipLaunchKernelGGL(k_mfcc, dim3(blocks), dim3(256), shm, s, fe, melw_n, wav, wav_off, frame_off, B,
                     total_frames, mfcc);
}

// -------------------------------------------------------------------- VAD
// The workgroup that finishes last (device-wide counter) also turns the voiced counts into the row
// offsets of the compacted feature matrix (select-voiced-frames bookkeeping), so no separate scan
// kernel is launched.  The counter is left at zero for the next launch.
__global__ __launch_bounds__(256) void k_vad(FbFrontendDev fe, const float *__restrict__ mfcc,
                                             const int *__restrict__ frame_off, int *__restrict__ vrank,
                                             int *__restrict__ tv, int B, int *__restrict__ counter,
                                             int *__restrict__ row_off) {
  const int b = blockIdx.x;
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  __shared__ double red[256];
  __shared__ float s_thr;
  __shared__ int s_run, s_wtot[4];
  double part = 0.0;
  for (int t = threadIdx.x; t < T; t += 256) part += (double)mfcc[(size_t)(base + t) * fe.nc];
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    s_thr = (float)(fe.vad_thr + fe.vad_mean_scale * red[0] / (double)T);
    s_run = 0;
  }
  __syncthreads();
  const float thr = s_thr;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + threadIdx.x;
    int v = 0;
    if (t < T) {
      int num = 0, den = 0;
      for (int t2 = t - fe.vad_ctx; t2 <= t + fe.vad_ctx; ++t2)
        if (t2 >= 0 && t2 < T) { ++den; if (mfcc[(size_t)(base + t2) * fe.nc] > thr) ++num; }
      v = ((float)num >= (float)den * fe.vad_prop) ? 1 : 0;
    }
    const unsigned long long bal = __ballot(v);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wtot[w] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; ++i) woff += s_wtot[i];
    if (t < T) vrank[base + t] = v ? (s_run + woff + pre) : -1;
    __syncthreads();
    if (threadIdx.x == 0) s_run += s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
    __syncthreads();
  }
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&tv[b], s_run, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    // (agent-scope store, agent-scope counter, agent-scope loads below: only this thread's store has to be complete
    //  before its increment -- no device-wide fences, see k_gmm_finalize_loss)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    s_last = (atomicAdd(counter, 1) == B - 1);
  }
  __syncthreads();
  if (!s_last) return;
  // exclusive scan of max(tv, 0): thread = contiguous slice, wave scan by shuffles, 4 wave totals
  const int per = (B + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(B, lo + per);
  int sum = 0;
  for (int i = lo; i < hi; ++i) {
    const int v = __hip_atomic_load(&tv[i], FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
    sum += v > 0 ? v : 0;
  }
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) s_wtot[w] = inc;
  __syncthreads();
  int run = inc - sum;
  for (int i = 0; i < w; ++i) run += s_wtot[i];
  if (threadIdx.x == 255) { row_off[B] = run + sum; *counter = 0; }
  for (int i = lo; i < hi; ++i) {
    const int v = __hip_atomic_load(&tv[i], FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
    row_off[i] = run;
    run += v > 0 ? v : 0;
  }
}
void fb_launch_vad(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                   int *vrank, int *tv, int *counter, int *row_off) {
  hipLaunchKernelGGL(k_vad, dim3(B), dim3(256), 0, s, fe, mfcc, frame_off, vrank, tv, B, counter, row_off);
}

// ------------------------------------------------------------------ compressed-feature round trip
// fb_frontend_cfg.compress_feats: steps/make_mfcc.sh (gmm_ubm_kaldiHelper.py:138-140) writes the MFCC matrix with
// `copy-feats --compress=true`, i.e. through Kaldi's lossy CompressedMatrix; every later stage reads that form.
// Same arithmetic as fbo_compress_roundtrip (oracle/fb_oracle.c), every operation an explicitly rounded
// intrinsic so that no fma contraction can change a code.  Bit-identical to the oracle on the same matrix.
__device__ __forceinline__ int fb_cm_to_u16(float minv, float range, float v) {
  float f = __fdiv_rn(__fsub_rn(v, minv), range);
  f = f > 1.0f ? 1.0f : f;
  f = f < 0.0f ? 0.0f : f;
  return (int)__dadd_rn((double)__fmul_rn(f, 65535.0f), 0.499);
}
__device__ __forceinline__ float fb_cm_from_u16(float minv, float range, int v) {
  return __fadd_rn(minv, __fmul_rn(__fmul_rn(range, 1.52590218966964e-05F), (float)v));
}
__device__ __forceinline__ int fb_cm_to_char(float p0, float p25, float p75, float p100, float v) {
  if (v < p25) {
    const float f = __fdiv_rn(__fsub_rn(v, p0), __fsub_rn(p25, p0));
    return min(max((int)__dadd_rn((double)__fmul_rn(f, 64.0f), 0.5), 0), 64);
  }
  if (v < p75) {
    const float f = __fdiv_rn(__fsub_rn(v, p25), __fsub_rn(p75, p25));
    return min(max(64 + (int)__dadd_rn((double)__fmul_rn(f, 128.0f), 0.5), 64), 192);
  }
  const float f = __fdiv_rn(__fsub_rn(v, p75), __fsub_rn(p100, p75));
  return min(max(192 + (int)__dadd_rn((double)__fmul_rn(f, 63.0f), 0.5), 192), 255);
}
__device__ __forceinline__ float fb_cm_from_char(float p0, float p25, float p75, float p100, int c) {
  if (c <= 64) return __double2float_rn(__dadd_rn((double)p0, __dmul_rn((double)__fmul_rn(__fsub_rn(p25, p0), (float)c), 1 / 64.0)));
  if (c <= 192)
    return __double2float_rn(__dadd_rn((double)p25, __dmul_rn((double)__fmul_rn(__fsub_rn(p75, p25), (float)(c - 64)), 1 / 128.0)));
  return __double2float_rn(__dadd_rn((double)p75, __dmul_rn((double)__fmul_rn(__fsub_rn(p100, p75), (float)(c - 192)), 1 / 63.0)));
}
__device__ __forceinline__ float fb_wave_min_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float fb_wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// workgroup = (utterance, four columns), wave = column.
//   1. the matrix's global header: minimum and maximum of the whole T x nc matrix, reduced by every workgroup of the
//      utterance for itself (coalesced, L2-resident; a separate launch for 51 pairs of numbers cost 9 us);
//   2. the order statistics 0, T/4, 3(T/4), T-1 of the wave's column (what Kaldi's CompressedMatrix takes with
//      std::nth_element: only their VALUES matter).  The two inner ones by bisection on the order-preserving integer
//      image of the floats: 32 rounds, each a wave-wide count of the elements below two candidate keys (compare +
//      ballot + scalar popcount, so the decision is wave-uniform) -- keys in registers for T <= 512, else staged in LDS
//      (lds_cap per wave), else re-read from global memory every round;
//   3. every element -> byte -> float, written to a SECOND matrix: the (utterance, column group) workgroups of an
//      utterance all reduce the header of step 1 from the whole matrix, so the input must stay as it is while any of
//      them runs (in place a workgroup could read columns another one had already rewritten -- the rounded maximum of a
//      compressed column is not always the original's -- and the header would depend on the block schedule).
// Round 2 ranked every element against every other straight from global memory: T dependent L2 round trips per 64
// elements, 207 us per NES batch -- the longest kernel of the reference-pipeline mode.
__device__ __forceinline__ unsigned fb_float_key(float v) {  // a < b  <=>  key(a) < key(b)  (no NaNs; -0 < +0)
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float fb_key_float(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}
#define FB_CM_REG 8  // keys per lane held in registers: columns of up to 512 frames
// The column header of a CompressedMatrix column whose order-preserving keys a wave holds in registers (lane l: elements
// l, 64 + l, ..; 0xffffffff past T): the quartile elements by bisection on the keys (32 rounds of compare + ballot +
// scalar popcount: the decisions are wave-uniform), then Kaldi's percentile values p0 < p25 < p75 < p100.
// clo / chi: the column's minimum / maximum.  Shared by k_feat_compress and k_vad_delta_cmvn.
__device__ __forceinline__ void fb_cm_column_header(const unsigned (&kr)[FB_CM_REG], int T, float minv, float range, float clo,
                                                    float chi, float &p0, float &p25, float &p75, float &p100) {
  const int q = T / 4;
  unsigned k25 = 0u, k75 = 0u;  // the element of rank r (0-based) is the largest key k with #{x : key(x) < k} <= r
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned c25 = k25 | (1u << bit), c75 = k75 | (1u << bit);
    int n25 = 0, n75 = 0;
#pragma unroll
    for (int r = 0; r < FB_CM_REG; ++r) {
      if (64 * r < T) {  // wave-uniform
        n25 += __popcll(__ballot(kr[r] < c25));
        n75 += __popcll(__ballot(kr[r] < c75));
      }
    }
    if (n25 <= q) k25 = c25;
    if (n75 <= 3 * q) k75 = c75;
  }
  const float v25 = fb_key_float(k25), v75 = fb_key_float(k75);
  const int u0 = min(fb_cm_to_u16(minv, range, clo), 65532);
  const int u25 = min(max(fb_cm_to_u16(minv, range, v25), u0 + 1), 65533);
  const int u75 = min(max(fb_cm_to_u16(minv, range, v75), u25 + 1), 65534);
  const int u100 = max(fb_cm_to_u16(minv, range, chi), u75 + 1);
  p0 = fb_cm_from_u16(minv, range, u0);
  p25 = fb_cm_from_u16(minv, range, u25);
  p75 = fb_cm_from_u16(minv, range, u75);
  p100 = fb_cm_from_u16(minv, range, u100);
}
__global__ __launch_bounds__(256) void k_feat_compress(const float *__restrict__ mfcc, float *__restrict__ out,
                                                       const int *__restrict__ frame_off, int nc, int lds_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_key[];  // [4 waves][lds_cap]
  __shared__ float s_lo[4], s_hi[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  if (T <= 0) return;
  float minv, maxv;
  {
    const float *m = mfcc + (size_t)base * nc;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < T * nc; i += 256) { const float v = m[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    lo = fb_wave_min_f(lo);
    hi = fb_wave_max_f(hi);
    if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; }
    __syncthreads();
    minv = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    maxv = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
  }
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + wv);
  if (c >= nc) return;
  const float *col = mfcc + (size_t)base * nc + c;
  float *ocol = out + (size_t)base * nc + c;
  if (maxv == minv) maxv = __fadd_rn(minv, __fadd_rn(1.0f, fabsf(minv)));
  const float range = __fsub_rn(maxv, minv);
  if (T <= 8) {  // kTwoByteAuto
    if (lane < T) ocol[(size_t)lane * nc] = fb_cm_from_u16(minv, range, fb_cm_to_u16(minv, range, col[(size_t)lane * nc]));
    return;
  }
  const int q = T / 4;
  const bool in_reg = T <= 64 * FB_CM_REG, in_lds = !in_reg && T <= lds_cap;
  unsigned *sk = s_key + (size_t)wv * lds_cap;
  unsigned kr[FB_CM_REG];
  float lo = INFINITY, hi = -INFINITY;
  if (in_reg) {
#pragma unroll
    for (int r = 0; r < FB_CM_REG; ++r) {
      const int i = 64 * r + lane;
      const float v = col[(size_t)min(i, T - 1) * nc];
      kr[r] = i < T ? fb_float_key(v) : 0xffffffffu;  // never below a candidate
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  } else {
    for (int i = lane; i < T; i += 64) {
      const float v = col[(size_t)i * nc];
      if (in_lds) sk[i] = fb_float_key(v);
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
    if (in_lds) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  lo = fb_wave_min_f(lo);
  hi = fb_wave_max_f(hi);
  // the element of rank r (0-based) is the largest key k with #{x : key(x) < k} <= r
  unsigned k25 = 0u, k75 = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned c25 = k25 | (1u << bit), c75 = k75 | (1u << bit);
    int n25 = 0, n75 = 0;
    if (in_reg) {
#pragma unroll
      for (int r = 0; r < FB_CM_REG; ++r) {
        if (64 * r < T) {  // wave-uniform
          n25 += __popcll(__ballot(kr[r] < c25));
          n75 += __popcll(__ballot(kr[r] < c75));
        }
      }
    } else {
      for (int i0 = 0; i0 < T; i0 += 64) {
        const int i = i0 + lane;
        const unsigned key = in_lds ? sk[min(i, T - 1)] : fb_float_key(col[(size_t)min(i, T - 1) * nc]);
        n25 += __popcll(__ballot(i < T && key < c25));
        n75 += __popcll(__ballot(i < T && key < c75));
      }
    }
    if (n25 <= q) k25 = c25;
    if (n75 <= 3 * q) k75 = c75;
  }
  const float v25 = fb_key_float(k25), v75 = fb_key_float(k75);
  const int u0 = min(fb_cm_to_u16(minv, range, lo), 65532);
  const int u25 = min(max(fb_cm_to_u16(minv, range, v25), u0 + 1), 65533);
  const int u75 = min(max(fb_cm_to_u16(minv, range, v75), u25 + 1), 65534);
  const int u100 = max(fb_cm_to_u16(minv, range, hi), u75 + 1);
  const float p0 = fb_cm_from_u16(minv, range, u0), p25 = fb_cm_from_u16(minv, range, u25),
              p75 = fb_cm_from_u16(minv, range, u75), p100 = fb_cm_from_u16(minv, range, u100);
  if (in_reg) {
#pragma unroll
    for (int r = 0; r < FB_CM_REG; ++r) {
      const int i = 64 * r + lane;
      if (i < T) ocol[(size_t)i * nc] = fb_cm_from_char(p0, p25, p75, p100, fb_cm_to_char(p0, p25, p75, p100, fb_key_float(kr[r])));
    }
  } else {
    for (int i = lane; i < T; i += 64) {
      ocol[(size_t)i * nc] = fb_cm_from_char(p0, p25, p75, p100, fb_cm_to_char(p0, p25, p75, p100, in_lds ? fb_key_float(sk[i]) : col[(size_t)i * nc]));
    }
  }
}
void fb_launch_feat_compress(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, float *out, const int *frame_off, int B,
                             int t_max) {
  const int cap = t_max <= 64 * FB_CM_REG ? 1 : std::min(t_max, 3840);  // keys per wave in LDS (4 x 15 KB: no opt-in needed)
  hipLaunchKernelGGL(k_feat_compress, dim3(B, (fe.nc + 3) / 4), dim3(256), sizeof(unsigned) * 4 * (size_t)cap, s, mfcc,
                     out, frame_off, fe.nc, cap);
}

// ------------------------------------------------------------------ deltas
// One workgroup per 32-frame chunk of one utterance.  Besides add-deltas it leaves the per-chunk
// column sums of the delta features (float64, fixed order) so that the whole-utterance CMVN mean
// needs no second pass over the features and no atomics.
#define FB_CHUNK 32
__global__ __launch_bounds__(256) void k_deltas(FbFrontendDev fe, const float *__restrict__ mfcc,
                                                const int *__restrict__ frame_off,
                                                const int *__restrict__ chunk_off, int B,
                                                float *__restrict__ dfeat, double *__restrict__ chunk_sum) {
  extern __shared__ float s_df[];  // [FB_CHUNK][dim]
  const int ch = blockIdx.x;
  const int b = fb_find_utt(chunk_off, B, ch);
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  const int t0 = (ch - chunk_off[b]) * FB_CHUNK;
  const int nt = min(FB_CHUNK, T - t0);
  const int nc = fe.nc, dim = fe.dim;
  const int maxlen = 2 * fe.order * fe.dwin + 1;
  for (int it = threadIdx.x; it < nt * nc; it += 256) {
    const int tl = it / nc, d = it - tl * nc, t = t0 + tl;
    for (int i = 0; i <= fe.order; ++i) {
      const double *sc = fe.dscale + (size_t)i * maxlen;
      const int off = i * fe.dwin;
      double acc = 0.0;
      for (int j = -off; j <= off; ++j) {
        int tt = t + j;
        tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
        const double sv = sc[j + off];
        if (sv != 0.0) acc = __dadd_rn(acc, __dmul_rn(sv, (double)mfcc[(size_t)(base + tt) * nc + d]));
      }
      const float o = (float)acc;
      s_df[tl * dim + i * nc + d] = o;
      dfeat[(size_t)(base + t) * dim + i * nc + d] = o;
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < dim; d += 256) {
    double acc = 0.0;
    for (int tl = 0; tl < nt; ++tl) acc += (double)s_df[tl * dim + d];
    chunk_sum[(size_t)ch * dim + d] = acc;
  }
}
void fb_launch_deltas(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off,
                      const int *chunk_off, int B, int total_chunks, float *dfeat, double *chunk_sum) {
  if (total_chunks <= 0) return;
  hipLaunchKernelGGL(k_deltas, dim3(total_chunks), dim3(256), sizeof(float) * FB_CHUNK * fe.dim, s, fe, mfcc,
                     frame_off, chunk_off, B, dfeat, chunk_sum);
}

// -------------------------------------------------------- CMVN + selection
// apply-cmvn-sliding --center=true --norm-vars=false + select-voiced-frames.
// Fast path (T <= cmn_window, every NES batch): every window is the whole utterance, so the
// mean is the sum of the chunk sums; one workgroup per 32-frame chunk.
__global__ __launch_bounds__(256) void k_cmvn(FbFrontendDev fe, const float *__restrict__ dfeat,
                                              const int *__restrict__ frame_off,
                                              const int *__restrict__ chunk_off,
                                              const double *__restrict__ chunk_sum,
                                              const int *__restrict__ vrank, const int *__restrict__ row_off,
                                              int B, float *__restrict__ feats) {
  extern __shared__ double s_sum[];  // [dim]
  const int ch = blockIdx.x, dim = fe.dim;
  const int b = fb_find_utt(chunk_off, B, ch);
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  if (T > fe.cmn_window) return;  // handled by k_cmvn_sliding
  const int c0 = chunk_off[b], c1 = chunk_off[b + 1];
  for (int d = threadIdx.x; d < dim; d += 256) {
    double acc = 0.0;
    for (int c = c0; c < c1; ++c) acc += chunk_sum[(size_t)c * dim + d];
    s_sum[d] = acc;
  }
  __syncthreads();
  const int t0 = (ch - c0) * FB_CHUNK;
  const int nt = min(FB_CHUNK, T - t0);
  const int rbase = row_off[b];
  const double alpha = (double)(float)(-1.0 / (double)T);
  for (int i = threadIdx.x; i < nt * dim; i += 256) {
    const int tl = i / dim, d = i - tl * dim;
    const int r = vrank[base + t0 + tl];
    if (r >= 0)
      feats[(size_t)(rbase + r) * dim + d] =
          (float)__dadd_rn((double)dfeat[(size_t)(base + t0 + tl) * dim + d], __dmul_rn(alpha, s_sum[d]));
  }
}
// General path (T > cmn_window): per-dimension running window sum in Kaldi's order, one
// workgroup per utterance (long enrolment / test utterances, never inside the NES loop).
__global__ __launch_bounds__(256) void k_cmvn_sliding(FbFrontendDev fe, const float *__restrict__ dfeat,
                                                      const int *__restrict__ frame_off,
                                                      const int *__restrict__ vrank,
                                                      const int *__restrict__ row_off,
                                                      float *__restrict__ feats) {
  const int b = blockIdx.x, dim = fe.dim;
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  const int rbase = row_off[b];
  const int Wn = fe.cmn_window;
  if (T <= Wn) return;
  for (int d = threadIdx.x; d < dim; d += 256) {
    double cur = 0.0;
    int lwb = 0, lwe = 0;
    for (int t = 0; t < T; ++t) {
      int wb = t - Wn / 2, we = wb + Wn;
      if (wb < 0) { we -= wb; wb = 0; }
      if (we > T) { wb -= (we - T); we = T; if (wb < 0) wb = 0; }
      for (; lwe < we; ++lwe) cur += (double)dfeat[(size_t)(base + lwe) * dim + d];
      for (; lwb < wb; ++lwb) cur -= (double)dfeat[(size_t)(base + lwb) * dim + d];
      const int r = vrank[base + t];
      if (r >= 0) {
        const double alpha = (double)(float)(-1.0 / (double)(we - wb));
        feats[(size_t)(rbase + r) * dim + d] =
            (float)__dadd_rn((double)dfeat[(size_t)(base + t) * dim + d], __dmul_rn(alpha, cur));
      }
    }
  }
}
// The CMVN column sum of an utterance's delta features, float64, in a FIXED blocked order: FB_CMVN_PARTS blocks of
// ceil(T / FB_CMVN_PARTS) frames, each summed frame by frame from zero, the block sums combined left to right.  (Kaldi's
// running window sum is the straight frame-by-frame order; float64 makes the difference ~1e-16 relative, and one order
// shared by every kernel below keeps them bit-identical to each other: k_vad_delta_cmvn_p splits an utterance over
// FB_CMVN_PARTS workgroups, one block each.)
#define FB_CMVN_PARTS 4
__device__ __forceinline__ double fb_cmvn_colsum(const float *__restrict__ s_df, int T, int dim, int d) {
  const int tq = (T + FB_CMVN_PARTS - 1) / FB_CMVN_PARTS;
  double tot = 0.0;
#pragma unroll
  for (int p = 0; p < FB_CMVN_PARTS; ++p) {
    const int t0 = min(T, p * tq), t1 = min(T, t0 + tq);
    double acc = 0.0;
    int t = t0;
    for (; t + 8 <= t1; t += 8) {  // the loads of 8 frames are issued before their (dependent) adds
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = s_df[(t + u) * dim + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)x[u];
    }
    for (; t < t1; ++t) acc += (double)s_df[t * dim + d];
    tot = p == 0 ? acc : tot + acc;
  }
  return tot;
}
// add-deltas | apply-cmvn-sliding | select-voiced-frames for utterances that fit the CMVN window
// (every NES batch): one workgroup per utterance keeps the MFCCs and the delta features in LDS, so
// the delta features never travel to HBM.  The mean is the float64 sum over the frames in order
// (Kaldi's running window sum when the window covers the whole utterance).
// ORDER/WIN > 0: delta options known at compile time (the recipe's 2/3 and Kaldi's default 2/2): the
// tap loops unroll, coefficients live in registers and all LDS loads of an output are issued before
// the dependent float64 adds.  ORDER < 0: any options.
template <int ORDER, int WIN>
__global__ __launch_bounds__(1024) void k_delta_cmvn(FbFrontendDev fe, const float *__restrict__ mfcc,
                                                     const int *__restrict__ frame_off,
                                                     const int *__restrict__ vrank,
                                                     const int *__restrict__ row_off, int t_cap,
                                                     float *__restrict__ feats) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  const int b = blockIdx.x, nc = fe.nc, dim = fe.dim, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  const int order = ORDER > 0 ? ORDER : fe.order, dwin = ORDER > 0 ? WIN : fe.dwin;
  const int maxlen = 2 * order * dwin + 1;
  double *s_sum = s_dyn;                                        // [dim]
  double *s_sc = s_sum + dim;                                   // [(order+1)][maxlen]
  const int ctx = order * dwin;
  float *s_mf = reinterpret_cast<float *>(s_sc + (order + 1) * maxlen);  // [ctx + T + ctx][nc], edges replicated
  float *s_df = s_mf + (size_t)(t_cap + 2 * ctx) * nc;          // [T][dim]
  int *s_vr = reinterpret_cast<int *>(s_df + (size_t)t_cap * dim);  // [T]
  for (int i = tid; i < (order + 1) * maxlen; i += 1024) s_sc[i] = fe.dscale[i];
  {  // MFCCs of the utterance: issue every load of this thread before the first LDS store
    constexpr int NL = 8;
    const float *src = mfcc + (size_t)base * nc;
    const int n = T * nc;
    for (int i0 = tid; i0 < n; i0 += 1024 * NL) {
      float v[NL];
#pragma unroll
      for (int u = 0; u < NL; ++u) v[u] = src[min(i0 + 1024 * u, n - 1)];
#pragma unroll
      for (int u = 0; u < NL; ++u) if (i0 + 1024 * u < n) s_mf[ctx * nc + i0 + 1024 * u] = v[u];
    }
    for (int i = tid; i < ctx * nc; i += 1024) {  // Kaldi clamps the frame index at both ends
      const int d = i % nc;
      s_mf[i] = src[d];
      s_mf[(ctx + T) * nc + i] = src[(size_t)(T - 1) * nc + d];
    }
  }
  for (int i = tid; i < T; i += 1024) s_vr[i] = vrank[base + i];
  __syncthreads();
  double creg[ORDER > 0 ? ORDER + 1 : 1][ORDER > 0 ? 2 * ORDER * WIN + 1 : 1];
  if constexpr (ORDER > 0) {
#pragma unroll
    for (int i = 0; i <= ORDER; ++i)
#pragma unroll
      for (int j = 0; j < 2 * ORDER * WIN + 1; ++j) creg[i][j] = s_sc[i * (2 * ORDER * WIN + 1) + j];
  }
  // ---- add-deltas: half-wave = frame, lane = coefficient (no divisions); float64 taps in order
  for (int d0 = 0; d0 < nc; d0 += 32) {
    const int d = d0 + (lane & 31);
    for (int t = 2 * w + (lane >> 5); t < T; t += 32) {
      if (d < nc) {
        if constexpr (ORDER > 0) {
#pragma unroll
          for (int i = 0; i <= ORDER; ++i) {
            constexpr int ML = 2 * ORDER * WIN + 1;
            const int off = i * WIN;
            float x[ML];
            const float *row = s_mf + (t + ctx - off) * nc + d;
#pragma unroll
            for (int j = 0; j < ML; ++j)
              if (j <= 2 * off) x[j] = row[j * nc];
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < ML; ++j)
              if (j <= 2 * off) acc = __dadd_rn(acc, __dmul_rn(creg[i][j], (double)x[j]));
            s_df[t * dim + i * nc + d] = (float)acc;
          }
        } else {
          for (int i = 0; i <= order; ++i) {
            const double *sc = s_sc + i * maxlen;
            const int off = i * dwin;
            double acc = 0.0;
            // (zero coefficients are not skipped as add-deltas does: adding an exact zero changes nothing)
            for (int j = 0; j <= 2 * off; ++j)
              acc = __dadd_rn(acc, __dmul_rn(sc[j], (double)s_mf[(t + ctx - off + j) * nc + d]));
            s_df[t * dim + i * nc + d] = (float)acc;
          }
        }
      }
    }
  }
  __syncthreads();
  if (tid < dim) s_sum[tid] = fb_cmvn_colsum(s_df, T, dim, tid);
  __syncthreads();
  // ---- CMVN + voiced-row compaction: wave = frame, lane = dimension
  const int rbase = row_off[b];
  const double alpha = (double)(float)(-1.0 / (double)T);
  for (int d = lane; d < dim; d += 64) {
    const double shift = __dmul_rn(alpha, s_sum[d]);
#pragma unroll 4
    for (int t = w; t < T; t += 16) {
      const int r = s_vr[t];
      if (r >= 0) feats[(size_t)(rbase + r) * dim + d] = (float)__dadd_rn((double)s_df[t * dim + d], shift);
    }
  }
}
// compute-vad-decision | add-deltas | apply-cmvn-sliding | select-voiced-frames in ONE launch (every NES batch).
// k_delta_cmvn with the VAD of k_vad in front -- the MFCCs of the utterance are in LDS anyway -- and the row offsets
// of the compacted feature matrix obtained without a scan kernel:
//   * workgroups take their utterance index from a ticket counter, so every utterance with a smaller index has
//     already been started by a running workgroup (no assumption about dispatch order);
//   * a workgroup publishes its voiced-frame count as (launch epoch << 32 | count) with an agent-scope release store
//     as soon as the VAD is done, computes deltas and CMVN sums, and only then -- usually without waiting -- reads the
//     counts of the utterances before it (acquire loads, spinning on a stale epoch) and adds them up;
//   * the workgroup that takes the last ticket resets the ticket counter; the one that holds utterance B - 1 also
//     writes the total (row_off[B], what the GMM kernel reads as its row count).
// Arithmetic, summation orders and results are those of k_vad + k_delta_cmvn, bit for bit.
template <int ORDER, int WIN>
__global__ __launch_bounds__(1024) void k_vad_delta_cmvn(FbFrontendDev fe, const float *mfcc,  // (no __restrict__: cm_out may be the same buffer)
                                                         const int *__restrict__ frame_off, int B, int t_cap,
                                                         unsigned epoch, int *__restrict__ ticket,
                                                         unsigned long long *__restrict__ pub, int *__restrict__ tv,
                                                         int *__restrict__ row_off, float *__restrict__ feats,
                                                         float *cm_out) {  // (may be mfcc itself)
  if (fe.stop && *fe.stop) return;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  __shared__ int s_b, s_run, s_wtot[16], s_rbase;
  __shared__ float s_thr, s_wlo[16], s_whi[16];
  const int nc = fe.nc, dim = fe.dim, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    const int tk = atomicAdd(ticket, 1);
    if (tk == B - 1) __hip_atomic_store(ticket, 0, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);  // nobody else will draw
    s_b = tk;
    s_run = 0;
  }
  __syncthreads();
  const int b = s_b;
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  const int order = ORDER > 0 ? ORDER : fe.order, dwin = ORDER > 0 ? WIN : fe.dwin;
  const int maxlen = 2 * order * dwin + 1;
  double *s_sum = s_dyn;                                        // [dim]
  double *s_sc = s_sum + dim;                                   // [(order+1)][maxlen]
  const int ctx = order * dwin;
  float *s_mf = reinterpret_cast<float *>(s_sc + (order + 1) * maxlen);  // [ctx + T + ctx][nc], edges replicated
  float *s_df = s_mf + (size_t)(t_cap + 2 * ctx) * nc;          // [T][dim]
  int *s_vr = reinterpret_cast<int *>(s_df + (size_t)t_cap * dim);  // [T]
  double *s_red = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(s_vr + t_cap) + 7) & ~(uintptr_t)7);  // [256]
  for (int i = tid; i < (order + 1) * maxlen; i += 1024) s_sc[i] = fe.dscale[i];
  {  // MFCCs of the utterance: issue every load of this thread before the first LDS store
    constexpr int NL = 8;
    const float *src = mfcc + (size_t)base * nc;
    const int n = T * nc;
    for (int i0 = tid; i0 < n; i0 += 1024 * NL) {
      float v[NL];
#pragma unroll
      for (int u = 0; u < NL; ++u) v[u] = src[min(i0 + 1024 * u, n - 1)];
#pragma unroll
      for (int u = 0; u < NL; ++u) if (i0 + 1024 * u < n) s_mf[ctx * nc + i0 + 1024 * u] = v[u];
    }
    for (int i = tid; i < ctx * nc; i += 1024) {  // Kaldi clamps the frame index at both ends
      const int d = i % nc;
      s_mf[i] = src[d];
      s_mf[(ctx + T) * nc + i] = src[(size_t)(T - 1) * nc + d];
    }
  }
  __syncthreads();
  // ---- cm_out != nullptr: Kaldi's CompressedMatrix round trip of the utterance's MFCC matrix (k_feat_compress's
  //      arithmetic on the copy in LDS: the launcher takes this path for T <= 64 FB_CM_REG), stored back for whoever
  //      reads the matrix later; the replicated edge rows are renewed from the compressed ones
  if (cm_out != nullptr) {
    float *m = s_mf + ctx * nc;  // [T][nc]
    float lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < T * nc; i += 1024) { const float v = m[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    lo = fb_wave_min_f(lo);
    hi = fb_wave_max_f(hi);
    if (lane == 0) { s_wlo[w] = lo; s_whi[w] = hi; }
    __syncthreads();
    float minv = s_wlo[0], maxv = s_whi[0];
    for (int i = 1; i < 16; ++i) { minv = fminf(minv, s_wlo[i]); maxv = fmaxf(maxv, s_whi[i]); }
    if (maxv == minv) maxv = __fadd_rn(minv, __fadd_rn(1.0f, fabsf(minv)));
    const float range = __fsub_rn(maxv, minv);
    if (T <= 8) {  // kTwoByteAuto
      for (int i = tid; i < T * nc; i += 1024) m[i] = fb_cm_from_u16(minv, range, fb_cm_to_u16(minv, range, m[i]));
    } else {
      for (int c = w; c < nc; c += 16) {  // wave = column; keys in registers
        unsigned kr[FB_CM_REG];
        float clo = INFINITY, chi = -INFINITY;
#pragma unroll
        for (int r = 0; r < FB_CM_REG; ++r) {
          const int i = 64 * r + lane;
          const float v = m[(size_t)min(i, T - 1) * nc + c];
          kr[r] = i < T ? fb_float_key(v) : 0xffffffffu;  // never below a candidate
          clo = fminf(clo, v);
          chi = fmaxf(chi, v);
        }
        clo = fb_wave_min_f(clo);
        chi = fb_wave_max_f(chi);
        float p0, p25, p75, p100;
        fb_cm_column_header(kr, T, minv, range, clo, chi, p0, p25, p75, p100);
#pragma unroll
        for (int r = 0; r < FB_CM_REG; ++r) {
          const int i = 64 * r + lane;
          if (i < T) m[(size_t)i * nc + c] = fb_cm_from_char(p0, p25, p75, p100, fb_cm_to_char(p0, p25, p75, p100, fb_key_float(kr[r])));
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < T * nc; i += 1024) cm_out[(size_t)base * nc + i] = m[i];
    for (int i = tid; i < ctx * nc; i += 1024) {
      const int d = i % nc;
      s_mf[i] = m[d];
      s_mf[(ctx + T) * nc + i] = m[(size_t)(T - 1) * nc + d];
    }
    __syncthreads();
  }
  // ---- VAD on the C0 column (k_vad's arithmetic: 256 strided float64 partial sums, binary tree; +-vad_ctx vote)
  const float *c0 = s_mf + ctx * nc;  // c0[t * nc]
  if (tid < 256) {
    double part = 0.0;
    for (int t = tid; t < T; t += 256) part += (double)c0[(size_t)t * nc];
    s_red[tid] = part;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) s_red[tid] += s_red[tid + o];
    __syncthreads();
  }
  if (tid == 0) s_thr = (float)(fe.vad_thr + fe.vad_mean_scale * s_red[0] / (double)T);
  __syncthreads();
  const float thr = s_thr;
  for (int t0 = 0; t0 < T; t0 += 1024) {
    const int t = t0 + tid;
    int v = 0;
    if (t < T) {
      int num = 0, den = 0;
      for (int t2 = t - fe.vad_ctx; t2 <= t + fe.vad_ctx; ++t2)
        if (t2 >= 0 && t2 < T) { ++den; if (c0[(size_t)t2 * nc] > thr) ++num; }
      v = ((float)num >= (float)den * fe.vad_prop) ? 1 : 0;
    }
    const unsigned long long bal = __ballot(v);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wtot[w] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; ++i) woff += s_wtot[i];
    if (t < T) s_vr[t] = v ? (s_run + woff + pre) : -1;
    __syncthreads();
    if (tid == 0) { int a = 0; for (int i = 0; i < 16; ++i) a += s_wtot[i]; s_run += a; }
    __syncthreads();
  }
  const int n_voiced = s_run;
  if (tid == 0) {
    tv[b] = n_voiced;
    __hip_atomic_store(&pub[b], ((unsigned long long)epoch << 32) | (unsigned)n_voiced, FB_XCH_ST,
                       __HIP_MEMORY_SCOPE_AGENT);   // (epoch and count in one word: nothing else to order, no fence)
  }
  double creg[ORDER > 0 ? ORDER + 1 : 1][ORDER > 0 ? 2 * ORDER * WIN + 1 : 1];
  if constexpr (ORDER > 0) {
#pragma unroll
    for (int i = 0; i <= ORDER; ++i)
#pragma unroll
      for (int j = 0; j < 2 * ORDER * WIN + 1; ++j) creg[i][j] = s_sc[i * (2 * ORDER * WIN + 1) + j];
  }
  // ---- add-deltas: thread = (frame, coefficient) pairs in flat order -- every lane busy (nc = 24 fills 3/4 of a
  //      half-wave), consecutive lanes on consecutive LDS words; float64 taps in order
  for (int idx = tid; idx < T * nc; idx += 1024) {
    const int t = idx / nc, d = idx - t * nc;
    if constexpr (ORDER > 0) {
#pragma unroll
      for (int i = 0; i <= ORDER; ++i) {
        constexpr int ML = 2 * ORDER * WIN + 1;
        const int off = i * WIN;
        float x[ML];
        const float *row = s_mf + (t + ctx - off) * nc + d;
#pragma unroll
        for (int j = 0; j < ML; ++j)
          if (j <= 2 * off) x[j] = row[j * nc];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < ML; ++j)
          if (j <= 2 * off) acc = __dadd_rn(acc, __dmul_rn(creg[i][j], (double)x[j]));
        s_df[t * dim + i * nc + d] = (float)acc;
      }
    } else {
      for (int i = 0; i <= order; ++i) {
        const double *sc = s_sc + i * maxlen;
        const int off = i * dwin;
        double acc = 0.0;
        for (int j = 0; j <= 2 * off; ++j)
          acc = __dadd_rn(acc, __dmul_rn(sc[j], (double)s_mf[(t + ctx - off + j) * nc + d]));
        s_df[t * dim + i * nc + d] = (float)acc;
      }
    }
  }
  __syncthreads();
  const double alpha = (double)(float)(-1.0 / (double)T);
  if (tid < dim) s_sum[tid] = __dmul_rn(alpha, fb_cmvn_colsum(s_df, T, dim, tid));  // the shift CMVN adds
  // ---- row offset: voiced counts of the utterances before this one (published above by their workgroups), read by
  //      the waves the sums leave idle (dim <= 128 here: waves 2 .. 15; otherwise by everybody, after the sums)
  {
    const int first = dim <= 128 ? 128 : 0, nth = 1024 - first;
    int mine = 0;
    if (tid >= first) {
      for (int i = tid - first; i < b; i += nth) {
        unsigned long long v;
        do {
          v = __hip_atomic_load(&pub[i], FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned)(v >> 32) != epoch);
        const int c = (int)(unsigned)v;
        mine += c > 0 ? c : 0;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) s_wtot[w] = mine;
  }
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int i = 0; i < 16; ++i) a += s_wtot[i];
    s_rbase = a;
    row_off[b] = a;
    if (b == B - 1) row_off[B] = a + (n_voiced > 0 ? n_voiced : 0);
  }
  __syncthreads();
  // ---- CMVN + voiced-row compaction: thread = (frame, dimension) pairs in flat order (dim = 72 leaves 8 of the 64
  //      lanes of a second pass busy otherwise); voiced frames follow each other, so do their output rows
  const int rbase = s_rbase;
  for (int idx = tid; idx < T * dim; idx += 1024) {
    const int t = idx / dim, d = idx - t * dim;
    const int r = s_vr[t];
    if (r >= 0) feats[(size_t)(rbase + r) * dim + d] = (float)__dadd_rn((double)s_df[idx], s_sum[d]);
  }
}
// k_vad_delta_cmvn with an utterance split over FB_CMVN_PARTS workgroups of 256 threads (round 4: one workgroup per
// utterance kept 51 of 256 compute units busy for 25 us per NES batch).  Part p owns the frames [p tq, (p + 1) tq), tq =
// ceil(T / FB_CMVN_PARTS):
//   * every part takes the whole C0 column (T floats) and repeats the VAD of the utterance -- mean, votes, ranks: a few
//     hundred operations -- so each knows all voiced ranks and the count; part 0 publishes the count (pub / tv);
//   * deltas of the own frames from the own MFCC rows +- the delta context (rows clamped at the utterance's ends, as
//     Kaldi clamps the frame index);
//   * the CMVN column sums need every frame: each part sums its own block (fb_cmvn_colsum's order), stores the dim
//     float64 block sums, polls the other parts' slots until they hold a sum and combines the blocks left to right --
//     the same float64 result in every part, bit for bit what the one-workgroup kernels compute;
//   * row offset from the published counts of the utterances before this one, voiced rows of the own frames written at
//     their final position.
// Workgroups draw (utterance, part) from a ticket: part p of utterance b has ticket FB_CMVN_PARTS b + p, so every
// workgroup with a smaller ticket is running or done.  A part also waits for LATER tickets -- the other parts of its own
// utterance --, which cannot deadlock: at any time at most one utterance is only partly started, every other started
// utterance has all its parts running and completes without anybody else's help, freeing its slots.
#ifdef FB_VAD_STAMP  // instrumented build (tools/profile/vad_instrumented.sh): where a workgroup of k_vad_delta_cmvn_p spends its time
__device__ unsigned long long g_vad_stamps[256 * 12];
extern "C" int fb_debug_vad_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vad_stamps), sizeof(g_vad_stamps)) == hipSuccess ? 0 : -1;
}
#define VD_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_vad_stamps[blockIdx.x * 12 + (k)] = wall_clock64(); } while (0)
#else
#define VD_STAMP(k) do { } while (0)
#endif
// NC > 0: the number of cepstra as a compile-time constant (the recipe's 24): the row strides of the delta taps, the block
// sums and the row writes become instruction immediates -- with one wave per SIMD every address instruction is exposed.
#define FB_VADP_THREADS 512  // (1024: rows written 1.6 -> 0.8 us, votes 1.1 -> 1.5 us behind the 16-wave barrier; 13.9 us against 14.3, within a box's noise on the iteration)
template <int ORDER, int WIN, int NC>
__global__ __launch_bounds__(FB_VADP_THREADS) void k_vad_delta_cmvn_p(FbFrontendDev fe, const float *__restrict__ mfcc,
                                                          const int *__restrict__ frame_off, int B, int t_cap,
                                                          unsigned epoch, int *__restrict__ ticket,
                                                          unsigned long long *__restrict__ pub, int *__restrict__ tv,
                                                          int *__restrict__ row_off, float *__restrict__ feats,
                                                          double *__restrict__ part_sum, unsigned slot_set) {
  if (fe.stop && *fe.stop) {
    // A launch queued behind the stopping iteration computes nothing, but it still owes the NEXT launch its sentinels:
    // the host counts this launch, so the next one polls the slot set the last REAL launch filled.  Workgroup i puts the
    // sentinel back into the i-th (utterance, part) slot of that set (any bijection does).
    unsigned long long *nxt = reinterpret_cast<unsigned long long *>(part_sum) + (size_t)((slot_set + 1u) & 1u) * B * FB_CMVN_PARTS * fe.dim;
    for (int d = threadIdx.x; d < fe.dim; d += FB_VADP_THREADS)
      __hip_atomic_store(&nxt[(size_t)blockIdx.x * fe.dim + d], FB_VAD_SENTINEL, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  VD_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  constexpr int NT = FB_VADP_THREADS, NW = NT / 64;  // two waves per SIMD: the float64 tap chains and LDS round trips of one hide behind the other's
  __shared__ int s_tk, s_wtot[NW], s_wt2[NW], s_rbase;
  __shared__ float s_thr;
  constexpr int NP = FB_CMVN_PARTS;
  const int nc = NC > 0 ? NC : fe.nc, dim = NC > 0 && ORDER > 0 ? NC * (ORDER + 1) : fe.dim, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  // (atomicInc wraps to zero behind the last ticket: nobody else will draw)
  if (tid == 0) {
    s_tk = (int)atomicInc(reinterpret_cast<unsigned *>(ticket), (unsigned)(B * NP - 1));
  }
  const int order = ORDER > 0 ? ORDER : fe.order, dwin = ORDER > 0 ? WIN : fe.dwin;
  const int maxlen = 2 * order * dwin + 1, ctx = order * dwin;
  const int tq_cap = (t_cap + NP - 1) / NP;
  double *s_sum = s_dyn;                                        // [dim]
  double *s_sc = s_sum + dim;                                   // [(order+1)][maxlen]
  double *s_red = s_sc + (order + 1) * maxlen;                  // [256]
  float *s_c0 = reinterpret_cast<float *>(s_red + 256);         // [t_cap] C0 of every frame of the utterance
  float *s_mf = s_c0 + ((t_cap + 1) & ~1);                      // [ctx + tq_cap + ctx][nc] own rows, edges clamped
  float *s_df = s_mf + (size_t)(tq_cap + 2 * ctx) * nc;         // [tq_cap][dim]
  int *s_vr = reinterpret_cast<int *>(s_df + (size_t)tq_cap * dim);  // [t_cap]
  const int n_sc = (order + 1) * maxlen;
  const double sc_own = fe.dscale[min(tid, n_sc - 1)];          // (in flight while the ticket is drawn)
  __syncthreads();
  VD_STAMP(1);
  const int b = s_tk / NP, part = s_tk - b * NP;
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  const int tq = (T + NP - 1) / NP;
  const int t0 = min(T, part * tq), t1 = min(T, t0 + tq), Tn = t1 - t0;   // own frames
  {
    // All of a thread's loads are issued before the first of them is waited for (unconditional loads on clamped
    // indices): a plain copy loop waits once per trip.  (Staging the rows of the GUESS ticket == blockIdx while the
    // atomic is in flight was tried: the 204 atomics race, almost no workgroup draws its own index, and staging twice
    // costs more than the round trip saved.)
    const float *src = mfcc + (size_t)base * nc;
    const int n = (Tn + 2 * ctx) * nc;
    constexpr int KC = 1, KM = 5;
    float c0v[KC], mv[KM];
#pragma unroll
    for (int k = 0; k < KC; ++k) c0v[k] = src[(size_t)min(tid + NT * k, T - 1) * nc];
    int r = tid / nc, d = tid - r * nc;                         // row / column of element tid + NT k, stepped without dividing
    const int dr = NT / nc, dd = NT - dr * nc;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const int tt = min(max(t0 - ctx + r, 0), T - 1);          // Kaldi clamps the frame index at both ends
      mv[k] = src[(size_t)tt * nc + d];                         // (behind the last element: a clamped row, never stored)
      r += dr;
      d += dd;
      if (d >= nc) { d -= nc; ++r; }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (tid + NT * k < T) s_c0[tid + NT * k] = c0v[k];
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (tid + NT * k < n) s_mf[tid + NT * k] = mv[k];
    for (int t = tid + NT * KC; t < T; t += NT) s_c0[t] = src[(size_t)t * nc];   // (longer than the recipe's shapes)
    for (int i = tid + NT * KM; i < n; i += NT) {
      const int r2 = i / nc, d2 = i - r2 * nc;
      const int tt = min(max(t0 - ctx + r2, 0), T - 1);
      s_mf[i] = src[(size_t)tt * nc + d2];
    }
    if (tid < n_sc) s_sc[tid] = sc_own;
    for (int i = tid + NT; i < n_sc; i += NT) s_sc[i] = fe.dscale[i];
  }
  __syncthreads();
  VD_STAMP(2);
  // ---- VAD of the whole utterance on the C0 column (k_vad's arithmetic: 256 strided float64 partial sums, binary tree)
  if (tid < 256) {
    double prt = 0.0;
    for (int t = tid; t < T; t += 256) prt += (double)s_c0[t];
    s_red[tid] = prt;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) s_red[tid] += s_red[tid + o];
    __syncthreads();
  }
  if (tid == 0) s_thr = (float)(fe.vad_thr + fe.vad_mean_scale * s_red[0] / (double)T);
  __syncthreads();
  const float thr = s_thr;
  VD_STAMP(3);
  // votes and voiced ranks, NT frames per trip (one for the recipe's utterances): a ballot per wave, ONE barrier, and
  // every thread adds up the wave counts itself
  int run = 0;
  for (int tb = 0; tb < T; tb += NT) {
    const int t = tb + tid;
    int v = 0;
    if (t < T) {
      int num = 0, den = 0;
      for (int t2 = t - fe.vad_ctx; t2 <= t + fe.vad_ctx; ++t2)
        if (t2 >= 0 && t2 < T) { ++den; if (s_c0[t2] > thr) ++num; }
      v = ((float)num >= (float)den * fe.vad_prop) ? 1 : 0;
    }
    const unsigned long long bal = __ballot(v);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wt2[w] = __popcll(bal);
    __syncthreads();
    int before = run;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int c = s_wt2[i];
      if (i == w && t < T) s_vr[t] = v ? before + pre : -1;
      before += c;
    }
    run = before;
    if (tb + NT < T) __syncthreads();
  }
  const int n_voiced = run;
  VD_STAMP(4);
  if (tid == 0 && part == 0) {
    tv[b] = n_voiced;
    __hip_atomic_store(&pub[b], ((unsigned long long)epoch << 32) | (unsigned)n_voiced, FB_XCH_ST,
                       __HIP_MEMORY_SCOPE_AGENT);   // (epoch and count in one word: nothing else to order)
  }
  double creg[ORDER > 0 ? ORDER + 1 : 1][ORDER > 0 ? 2 * ORDER * WIN + 1 : 1];
  if constexpr (ORDER > 0) {
#pragma unroll
    for (int i = 0; i <= ORDER; ++i)
#pragma unroll
      for (int j = 0; j < 2 * ORDER * WIN + 1; ++j) creg[i][j] = s_sc[i * (2 * ORDER * WIN + 1) + j];
  }
  // ---- add-deltas of the own frames: thread = (frame, coefficient) pairs in flat order; float64 taps in order
  int t = tid / nc, d = tid - t * nc;                           // t: own frame index; its row in s_mf is t + ctx
  const int t_step = NT / nc, d_step = NT - t_step * nc;        // (frame, coefficient) of idx + NT without dividing
  for (int idx = tid; idx < Tn * nc; idx += NT, t += t_step, d += d_step) {
    if (d >= nc) { d -= nc; ++t; }
    if constexpr (ORDER > 0) {
#pragma unroll
      for (int i = 0; i <= ORDER; ++i) {
        constexpr int ML = 2 * ORDER * WIN + 1;
        const int off = i * WIN;
        float x[ML];
        const float *row = s_mf + (t + ctx - off) * nc + d;
#pragma unroll
        for (int j = 0; j < ML; ++j)
          if (j <= 2 * off) x[j] = row[j * nc];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < ML; ++j)
          if (j <= 2 * off) acc = __dadd_rn(acc, __dmul_rn(creg[i][j], (double)x[j]));
        s_df[t * dim + i * nc + d] = (float)acc;
      }
    } else {
      for (int i = 0; i <= order; ++i) {
        const double *sc = s_sc + i * maxlen;
        const int off = i * dwin;
        double acc = 0.0;
        for (int j = 0; j <= 2 * off; ++j)
          acc = __dadd_rn(acc, __dmul_rn(sc[j], (double)s_mf[(t + ctx - off + j) * nc + d]));
        s_df[t * dim + i * nc + d] = (float)acc;
      }
    }
  }
  __syncthreads();
  VD_STAMP(5);
  // ---- this part's block of the CMVN column sums (frame by frame from zero), exchanged with the other parts.
  //      No flags and no fences: a block sum is ONE 64-bit word, stored and polled with relaxed agent-scope atomics; a
  //      slot that still holds the sentinel (a NaN no sum can be: a NaN sum is stored as another NaN) has not been
  //      written yet.  Two slot sets alternate with the launches of THIS kernel (slot_set: the host's count of them --
  //      not the epoch of `pub`, which k_vad_delta_cmvn's launches advance as well); every part puts the sentinel back
  //      into its slots of the OTHER set, which the previous launch used and the next one will.  (Device-wide release / acquire fences -- L2
  //      write-back and invalidate on a multi-XCD part -- cost this kernel 17 us when every thread issued one, and
  //      still ~2 us per workgroup with one release store + one acquire fence.)
  const double alpha = (double)(float)(-1.0 / (double)T);
  const unsigned long long SENT = FB_VAD_SENTINEL;
  unsigned long long *cur = reinterpret_cast<unsigned long long *>(part_sum) + (size_t)(slot_set & 1u) * B * NP * dim;
  unsigned long long *nxt = reinterpret_cast<unsigned long long *>(part_sum) + (size_t)((slot_set + 1u) & 1u) * B * NP * dim;
  double own = 0.0;
  for (int d = tid; d < dim; d += NT) {   // (dim <= 256: at most one dimension per thread)
    // frame by frame from zero.  (Requesting the next eight values before these eight are added, with a branch-free
    // tail, was tried twice: 1.3 -> 2.0 us -- the selects and clamps cost this single wave more than the round trips.)
    double acc = 0.0;
    int t = 0;
    for (; t + 8 <= Tn; t += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = s_df[(t + u) * dim + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)x[u];
    }
    for (; t < Tn; ++t) acc += (double)s_df[t * dim + d];
    own = acc;
    unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
    if (acc != acc) bits = 0x7ff8000000000001ull;
    __hip_atomic_store(&cur[((size_t)b * NP + part) * dim + d], bits, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&nxt[((size_t)b * NP + part) * dim + d], SENT, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
  }
  VD_STAMP(6);
  // ---- row offset: voiced counts of the utterances before this one (the count travels IN the polled word)
  {
    int mine = 0;
    for (int i = tid; i < b; i += NT) {
      unsigned long long v;
      do {
        v = __hip_atomic_load(&pub[i], FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
      } while ((unsigned)(v >> 32) != epoch);
      const int c = (int)(unsigned)v;
      mine += c > 0 ? c : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) s_wtot[w] = mine;
  }
  VD_STAMP(7);
  for (int d = tid; d < dim; d += NT) {
    double tot = 0.0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      double v = own;
      if (p != part) {
        unsigned long long bits;
        do {
          bits = __hip_atomic_load(&cur[((size_t)b * NP + p) * dim + d], FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
        } while (bits == SENT);
        v = __longlong_as_double((long long)bits);
      }
      tot = p == 0 ? v : tot + v;
    }
    s_sum[d] = __dmul_rn(alpha, tot);  // the shift CMVN adds
  }
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int i = 0; i < NW; ++i) a += s_wtot[i];
    s_rbase = a;
    if (part == 0) {
      row_off[b] = a;
      if (b == B - 1) row_off[B] = a + (n_voiced > 0 ? n_voiced : 0);
    }
  }
  __syncthreads();
  VD_STAMP(8);
  // ---- CMVN + voiced-row compaction of the own frames: thread = (frame, dimension) pairs in flat order
  const int rbase = s_rbase;
  int tw = tid / dim, dw = tid - tw * dim;
  const int tw_step = NT / dim, dw_step = NT - tw_step * dim;
  for (int idx = tid; idx < Tn * dim; idx += NT, tw += tw_step, dw += dw_step) {
    if (dw >= dim) { dw -= dim; ++tw; }
    const int r = s_vr[t0 + tw];
    if (r >= 0) feats[(size_t)(rbase + r) * dim + dw] = (float)__dadd_rn((double)s_df[idx], s_sum[dw]);
  }
  VD_STAMP(9);
}
size_t fb_vad_delta_cmvn_p_lds_bytes(const FbFrontendDev &fe, int t_cap) {
  const int maxlen = 2 * fe.order * fe.dwin + 1, tq_cap = (t_cap + FB_CMVN_PARTS - 1) / FB_CMVN_PARTS;
  return sizeof(double) * (size_t)(fe.dim + (fe.order + 1) * maxlen + 256) +
         sizeof(float) * ((size_t)((t_cap + 1) & ~1) + (size_t)(tq_cap + 2 * fe.order * fe.dwin) * fe.nc + (size_t)tq_cap * fe.dim) +
         sizeof(int) * (size_t)t_cap + 16;
}
// part_sum: two slot sets of B x FB_CMVN_PARTS x dim 64-bit words, every word FB_VAD_SENTINEL before the first launch;
// slot_set: the number of launches of this kernel on the buffer since then (a launch that finds the attack's stop flag
// raised counts: it restores the sentinels it owes).  Returns false when the batch does not qualify.
size_t fb_vad_parts_doubles(const FbFrontendDev &fe, int B) { return (size_t)2 * B * FB_CMVN_PARTS * fe.dim; }
bool fb_launch_vad_delta_cmvn_p(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                                int t_max, unsigned epoch, int *ticket, unsigned long long *pub, int *tv, int *row_off,
                                float *feats, double *part_sum, unsigned slot_set, bool spread) {
  if (B <= 0) return true;
  if (t_max > fe.cmn_window || fe.dim > 256) return false;
  size_t shm = fb_vad_delta_cmvn_p_lds_bytes(fe, t_max);
  if (shm > 64 * 1024) return false;
  // One workgroup per compute unit while the grid fits the chip: the dispatcher otherwise stacks the four parts of an
  // utterance on ONE unit (35 KB of LDS and 256 threads each fit four times) and the split buys nothing -- measured:
  // 39.6 us stacked against 24.7 us for the one-workgroup kernel.  Asking for more than half of a unit's LDS keeps
  // them apart.
  // ... when the attack has the GPU to itself (spread).  With three or more attacks in flight the padding is what hurts:
  // 82 KB keep the workgroup off every CU a k_gmm_fx2w workgroup of another attack (101 KB) sits on; at its own 35 KB it
  // runs beside them: 11.66 -> 12.08 k it/s (tools/profile/r05_stack.sh; padded, the fused front-end LOSES 1 % there).
  if (spread && B * FB_CMVN_PARTS <= 256 && getenv("FB_VADP_STACK") == nullptr) {  // (FB_VADP_STACK=1: never pad -- experiments)
    static std::atomic<unsigned long long> optin{0};
    unsigned long long bit = 0;
    bool ok = true;
    if (fb_device_needs_optin(optin, &bit)) {
      const void *fns[] = {reinterpret_cast<const void *>(k_vad_delta_cmvn_p<2, 3, 24>), reinterpret_cast<const void *>(k_vad_delta_cmvn_p<2, 3, 0>),
                           reinterpret_cast<const void *>(k_vad_delta_cmvn_p<2, 2, 0>), reinterpret_cast<const void *>(k_vad_delta_cmvn_p<-1, 0, 0>)};
      for (const void *fn : fns)
        ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
      if (ok) optin.fetch_or(bit, std::memory_order_release);
    }
    if (ok) shm = std::max(shm, (size_t)82 * 1024);
  }
  const dim3 grid((unsigned)(B * FB_CMVN_PARTS)), blk(FB_VADP_THREADS);
  if (fe.order == 2 && fe.dwin == 3 && fe.nc == 24 && fe.dim == 72)
    hipLaunchKernelGGL((k_vad_delta_cmvn_p<2, 3, 24>), grid, blk, shm, s, fe, mfcc, frame_off, B, t_max, epoch, ticket, pub, tv, row_off, feats, part_sum, slot_set);
  else if (fe.order == 2 && fe.dwin == 3)
    hipLaunchKernelGGL((k_vad_delta_cmvn_p<2, 3, 0>), grid, blk, shm, s, fe, mfcc, frame_off, B, t_max, epoch, ticket, pub, tv, row_off, feats, part_sum, slot_set);
  else if (fe.order == 2 && fe.dwin == 2)
    hipLaunchKernelGGL((k_vad_delta_cmvn_p<2, 2, 0>), grid, blk, shm, s, fe, mfcc, frame_off, B, t_max, epoch, ticket, pub, tv, row_off, feats, part_sum, slot_set);
  else
    hipLaunchKernelGGL((k_vad_delta_cmvn_p<-1, 0, 0>), grid, blk, shm, s, fe, mfcc, frame_off, B, t_max, epoch, ticket, pub, tv, row_off, feats, part_sum, slot_set);
  return true;
}

size_t fb_delta_cmvn_lds_bytes(const FbFrontendDev &fe, int t_cap) {
  const int maxlen = 2 * fe.order * fe.dwin + 1;
  return sizeof(double) * (size_t)(fe.dim + (fe.order + 1) * maxlen) +
         sizeof(float) * ((size_t)(t_cap + 2 * fe.order * fe.dwin) * fe.nc + (size_t)t_cap * fe.dim) +
         sizeof(int) * (size_t)t_cap + 16;
}
// returns false when the batch does not qualify (an utterance longer than the CMVN window or than
// the LDS allows): the caller then runs fb_launch_deltas + fb_launch_cmvn
bool fb_launch_delta_cmvn(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off,
                          const int *vrank, const int *row_off, int B, int t_max, float *feats) {
  if (B <= 0) return true;
  if (t_max > fe.cmn_window) return false;
  const size_t shm = fb_delta_cmvn_lds_bytes(fe, t_max);
  if (shm > 150 * 1024) return false;
  static std::atomic<unsigned long long> optin{0};  // raise the dynamic-LDS limit once per device
  unsigned long long bit = 0;
  if (fb_device_needs_optin(optin, &bit)) {
    const void *fns[] = {reinterpret_cast<const void *>(k_delta_cmvn<2, 3>), reinterpret_cast<const void *>(k_delta_cmvn<2, 2>),
                         reinterpret_cast<const void *>(k_delta_cmvn<-1, 0>)};
    for (const void *fn : fns)
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return false;
    optin.fetch_or(bit, std::memory_order_release);
  }
  if (fe.order == 2 && fe.dwin == 3)
    hipLaunchKernelGGL((k_delta_cmvn<2, 3>), dim3(B), dim3(1024), shm, s, fe, mfcc, frame_off, vrank, row_off, t_max, feats);
  else if (fe.order == 2 && fe.dwin == 2)
    hipLaunchKernelGGL((k_delta_cmvn<2, 2>), dim3(B), dim3(1024), shm, s, fe, mfcc, frame_off, vrank, row_off, t_max, feats);
  else
    hipLaunchKernelGGL((k_delta_cmvn<-1, 0>), dim3(B), dim3(1024), shm, s, fe, mfcc, frame_off, vrank, row_off, t_max, feats);
  return true;
}

// true when k_vad_delta_cmvn can take the CompressedMatrix round trip of a batch whose longest utterance has t_max frames
bool fb_vad_delta_cmvn_compresses(int t_max) { return t_max <= 64 * FB_CM_REG; }
// returns false when the batch does not qualify (as fb_launch_delta_cmvn): the caller then runs fb_launch_vad and
// the separate delta / CMVN kernels.  ticket: one int, zero before the first launch; pub: B x 8 bytes; epoch: a
// number no earlier launch on these buffers used (the engine counts launches).
bool fb_launch_vad_delta_cmvn(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                              int t_max, unsigned epoch, int *ticket, unsigned long long *pub, int *tv, int *row_off,
                              float *feats, float *cm_out) {
  if (B <= 0) return true;
  if (t_max > fe.cmn_window) return false;
  const size_t shm = fb_delta_cmvn_lds_bytes(fe, t_max) + 16 + sizeof(double) * 256;
  if (shm > 150 * 1024) return false;
  static std::atomic<unsigned long long> optin{0};
  unsigned long long bit = 0;
  if (fb_device_needs_optin(optin, &bit)) {
    const void *fns[] = {reinterpret_cast<const void *>(k_vad_delta_cmvn<2, 3>), reinterpret_cast<const void *>(k_vad_delta_cmvn<2, 2>),
                         reinterpret_cast<const void *>(k_vad_delta_cmvn<-1, 0>)};
    for (const void *fn : fns)
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return false;
    optin.fetch_or(bit, std::memory_order_release);
  }
  if (fe.order == 2 && fe.dwin == 3)
    hipLaunchKernelGGL((k_vad_delta_cmvn<2, 3>), dim3(B), dim3(1024), shm, s, fe, mfcc, frame_off, B, t_max, epoch, ticket, pub, tv, row_off, feats, cm_out);
  else if (fe.order == 2 && fe.dwin == 2)
    hipLaunchKernelGGL((k_vad_delta_cmvn<2, 2>), dim3(B), dim3(1024), shm, s, fe, mfcc, frame_off, B, t_max, epoch, ticket, pub, tv, row_off, feats, cm_out);
  else
    hipLaunchKernelGGL((k_vad_delta_cmvn<-1, 0>), dim3(B), dim3(1024), shm, s, fe, mfcc, frame_off, B, t_max, epoch, ticket, pub, tv, row_off, feats, cm_out);
  return true;
}

void fb_launch_cmvn(hipStream_t s, const FbFrontendDev &fe, const float *dfeat, const int *frame_off,
                    const int *chunk_off, const double *chunk_sum, const int *vrank, const int *row_off, int B,
                    int total_chunks, bool any_long, float *feats) {
  if (total_chunks <= 0) return;
  hipLaunchKernelGGL(k_cmvn, dim3(total_chunks), dim3(256), sizeof(double) * fe.dim, s, fe, dfeat, frame_off,
                     chunk_off, chunk_sum, vrank, row_off, B, feats);
  if (any_long)
    hipLaunchKernelGGL(k_cmvn_sliding, dim3(B), dim3(256), 0, s, fe, dfeat, frame_off, vrank, row_off, feats);
}
