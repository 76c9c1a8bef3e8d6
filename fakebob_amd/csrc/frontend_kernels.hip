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

// ------------------------------------------------------------------- MFCC
// One wave per frame, 4 frames per 256-thread block.  LDS per wave:
//   A[P] doubles (P/2 complex), Bf[P] doubles, PW[P/2+1] doubles.
#define FB_MFCC_MAXI 8  // samples per lane: frame_length <= 512

__global__ __launch_bounds__(256) void k_mfcc(FbFrontendDev fe, const int16_t *__restrict__ wav,
                                              const int64_t *__restrict__ wav_off,
                                              const int *__restrict__ frame_off, int B, int total_frames,
                                              float *__restrict__ mfcc) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int P = fe.P, Nc = P >> 1, L = fe.L;
  const int per_wave = 2 * P + (Nc + 8);
  double *A = smem + (size_t)w * per_wave;
  double *Bf = A + P;
  double *PW = Bf + P;
  int f = blockIdx.x * 4 + w;
  const bool valid = f < total_frames;
  if (!valid) f = total_frames - 1;
  const int b = fb_find_utt(frame_off, B, f);
  const int t = f - frame_off[b];
  const int64_t n = wav_off[b + 1] - wav_off[b];
  const int16_t *wv = wav + wav_off[b];
  const int64_t start = fe.snip_edges ? (int64_t)t * fe.shift : (int64_t)t * fe.shift + fe.shift / 2 - L / 2;

  // ---- load, DC removal, raw energy
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
  __syncthreads();
  // ---- pre-emphasis + window, packed as P/2 complex points (re=y[2k], im=y[2k+1])
  double en2 = 0.0;
#pragma unroll
  for (int i = 0; i < FB_MFCC_MAXI; ++i) {
    const int s = lane + 64 * i;
    if (s < P) {
      double y = 0.0;
      if (s < L) {
        const double prev = Bf[s > 0 ? s - 1 : 0];
        y = (xs[i] - fe.preemph * prev) * fe.window[s];
        en2 = fma(y, y, en2);
      }
      A[s] = y;
    }
  }
  if (!fe.raw_energy) energy = fb_wave_sum(en2);
  double log_energy = log(energy > (double)FLT_EPSILON ? energy : (double)FLT_EPSILON);
  if (log_energy < fe.log_energy_floor) log_energy = fe.log_energy_floor;
  __syncthreads();
  // ---- Stockham radix-2 complex FFT of size Nc
  double2 *src = reinterpret_cast<double2 *>(A), *dst = reinterpret_cast<double2 *>(Bf);
  const double2 *tw = reinterpret_cast<const double2 *>(fe.tw_half);
  for (int Ns = 1; Ns < Nc; Ns <<= 1) {
    const int tstep = Nc / (2 * Ns);
    for (int j = lane; j < Nc / 2; j += 64) {
      const int k = j & (Ns - 1);
      const double2 wv2 = tw[k * tstep];
      const double2 v0 = src[j], x1 = src[j + Nc / 2];
      double2 v1;
      v1.x = x1.x * wv2.x - x1.y * wv2.y;
      v1.y = x1.x * wv2.y + x1.y * wv2.x;
      const int idx = ((j - k) << 1) + k;
      dst[idx] = make_double2(v0.x + v1.x, v0.y + v1.y);
      dst[idx + Ns] = make_double2(v0.x - v1.x, v0.y - v1.y);
    }
    __syncthreads();
    double2 *tmp = src; src = dst; dst = tmp;
  }
  // ---- real-FFT unpack + power spectrum, bins 0..Nc
  const double2 *twf = reinterpret_cast<const double2 *>(fe.tw_full);
  for (int k = lane; k <= Nc; k += 64) {
    const double2 zk = src[k & (Nc - 1)];
    const double2 zr = src[(Nc - k) & (Nc - 1)];
    const double er = 0.5 * (zk.x + zr.x), ei = 0.5 * (zk.y - zr.y);  // E = (Zk + conj(Zr))/2
    const double dr = zk.x - zr.x, di = zk.y + zr.y;                  // d = Zk - conj(Zr)
    const double orr = 0.5 * di, oi = -0.5 * dr;                      // O = -i/2 * d
    const double2 wk = twf[k];
    const double xr = er + (wk.x * orr - wk.y * oi);
    const double xi = ei + (wk.x * oi + wk.y * orr);
    PW[k] = xr * xr + xi * xi;
  }
  __syncthreads();
  // ---- mel filterbank + log   (LM aliases dst)
  double *LM = reinterpret_cast<double *>(dst);
  for (int m = lane; m < fe.nb; m += 64) {
    const double *wm = fe.mel_w + fe.mel_off[m];
    const int first = fe.mel_first[m], len = fe.mel_len[m];
    double e = 0.0;
    for (int i = 0; i < len; ++i) e += wm[i] * PW[first + i];
    if (e < (double)FLT_EPSILON) e = (double)FLT_EPSILON;
    LM[m] = log(e);
  }
  __syncthreads();
  // ---- DCT-II, lifter, C0 <- log energy
  for (int c = lane; c < fe.nc; c += 64) {
    const double *dr = fe.dct + (size_t)c * fe.nb;
    double acc = 0.0;
    for (int m = 0; m < fe.nb; ++m) acc += dr[m] * LM[m];
    acc *= fe.lifter[c];
    float o = (float)acc;
    if (c == 0 && fe.use_energy) o = (float)log_energy;
    if (valid) mfcc[(size_t)f * fe.nc + c] = o;
  }
}

void fb_launch_mfcc(hipStream_t s, const FbFrontendDev &fe, const int16_t *wav, const int64_t *wav_off,
                    const int *frame_off, int B, int total_frames, float *mfcc) {
  if (total_frames <= 0) return;
  const int per_wave = 2 * fe.P + (fe.P / 2 + 8);
  size_t shm = sizeof(double) * (size_t)per_wave * 4;
  hipLaunchKernelGGL(k_mfcc, dim3((total_frames + 3) / 4), dim3(256), shm, s, fe, wav, wav_off, frame_off, B,
                     total_frames, mfcc);
}

// -------------------------------------------------------------------- VAD
__global__ __launch_bounds__(256) void k_vad(FbFrontendDev fe, const float *__restrict__ mfcc,
                                             const int *__restrict__ frame_off, int *__restrict__ vrank,
                                             int *__restrict__ tv) {
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
  if (threadIdx.x == 0) tv[b] = s_run;
}
void fb_launch_vad(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                   int *vrank, int *tv) {
  hipLaunchKernelGGL(k_vad, dim3(B), dim3(256), 0, s, fe, mfcc, frame_off, vrank, tv);
}

__global__ __launch_bounds__(256) void k_rowscan(const int *__restrict__ tv, int B, int *__restrict__ row_off) {
  // single block; B is small (utterances per batch)
  __shared__ int s_part[256];
  const int per = (B + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(B, lo + per);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += tv[i] > 0 ? tv[i] : 0;
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { int v = s_part[i]; s_part[i] = run; run += v; }
    row_off[B] = run;
  }
  __syncthreads();
  int run = s_part[threadIdx.x];
  for (int i = lo; i < hi; ++i) { row_off[i] = run; run += tv[i] > 0 ? tv[i] : 0; }
}
void fb_launch_rowscan(hipStream_t s, const int *tv, int B, int *row_off) {
  hipLaunchKernelGGL(k_rowscan, dim3(1), dim3(256), 0, s, tv, B, row_off);
}

// ------------------------------------------------------------------ deltas
__global__ __launch_bounds__(256) void k_deltas(FbFrontendDev fe, const float *__restrict__ mfcc,
                                                const int *__restrict__ frame_off, int B, int total_frames,
                                                float *__restrict__ dfeat) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int f = (int)(gid / fe.nc), d = (int)(gid % fe.nc);
  if (f >= total_frames) return;
  const int b = fb_find_utt(frame_off, B, f);
  const int base = frame_off[b], T = frame_off[b + 1] - base, t = f - base;
  const int maxlen = 2 * fe.order * fe.dwin + 1;
  for (int i = 0; i <= fe.order; ++i) {
    const double *sc = fe.dscale + (size_t)i * maxlen;
    const int off = i * fe.dwin;
    double acc = 0.0;
    for (int j = -off; j <= off; ++j) {
      int tt = t + j;
      tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
      const double sv = sc[j + off];
      if (sv != 0.0) acc = __dadd_rn(acc, __dmul_rn(sv, (double)mfcc[(size_t)(base + tt) * fe.nc + d]));
    }
    dfeat[(size_t)f * fe.dim + i * fe.nc + d] = (float)acc;
  }
}
void fb_launch_deltas(hipStream_t s, const FbFrontendDev &fe, const float *mfcc, const int *frame_off, int B,
                      int total_frames, float *dfeat) {
  int64_t n = (int64_t)total_frames * fe.nc;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_deltas, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fe, mfcc, frame_off, B,
                     total_frames, dfeat);
}

// -------------------------------------------------------- CMVN + selection
// One block per utterance.  T <= cmn_window: every window is the whole utterance
// (column mean); otherwise a per-dimension running window sum in Kaldi's order.
__global__ __launch_bounds__(256) void k_cmvn(FbFrontendDev fe, const float *__restrict__ dfeat,
                                              const int *__restrict__ frame_off,
                                              const int *__restrict__ vrank, const int *__restrict__ row_off,
                                              float *__restrict__ feats) {
  extern __shared__ double s_sum[];  // [groups][dim]
  const int b = blockIdx.x, dim = fe.dim;
  const int base = frame_off[b], T = frame_off[b + 1] - base;
  const int rbase = row_off[b];
  const int Wn = fe.cmn_window;
  if (T <= 0) return;
  if (T <= Wn) {
    const int groups = 256 / dim > 0 ? 256 / dim : 1;
    const int g = threadIdx.x / dim, d = threadIdx.x % dim;
    if (g < groups) {
      const int per = (T + groups - 1) / groups;
      const int lo = g * per, hi = min(T, lo + per);
      double acc = 0.0;
      for (int t = lo; t < hi; ++t) acc += (double)dfeat[(size_t)(base + t) * dim + d];
      s_sum[g * dim + d] = acc;
    }
    __syncthreads();
    if (threadIdx.x < dim) {
      double acc = 0.0;
      for (int g2 = 0; g2 < groups; ++g2) acc += s_sum[g2 * dim + threadIdx.x];
      s_sum[threadIdx.x] = acc;
    }
    __syncthreads();
    const double alpha = (double)(float)(-1.0 / (double)T);
    for (int i = threadIdx.x; i < T * dim; i += 256) {
      const int t = i / dim, d2 = i - t * dim;
      const int r = vrank[base + t];
      if (r >= 0)
        feats[(size_t)(rbase + r) * dim + d2] =
            (float)__dadd_rn((double)dfeat[(size_t)(base + t) * dim + d2], __dmul_rn(alpha, s_sum[d2]));
    }
  } else {
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
}
void fb_launch_cmvn(hipStream_t s, const FbFrontendDev &fe, const float *dfeat, const int *frame_off,
                    const int *vrank, const int *row_off, int B, float *feats) {
  const int groups = 256 / fe.dim > 0 ? 256 / fe.dim : 1;
  size_t shm = sizeof(double) * (size_t)groups * fe.dim;
  hipLaunchKernelGGL(k_cmvn, dim3(B), dim3(256), shm, s, fe, dfeat, frame_off, vrank, row_off, feats);
}
