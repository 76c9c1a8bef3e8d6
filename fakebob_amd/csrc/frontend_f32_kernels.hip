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
//   mel_m = the filter's weights in chunks of 12: every chunk an in-order chain fmaf(w_m[i], pow[first_m + i], .) from 0,
//           the chunk sums added left to right; log as (float)log_f64((double)max(mel, FLT_EPSILON))
//   c_k   = (A + B) * lifter[k], A / B = in-order chains fmaf(dct[k][m], logmel[m], .) over m < ceil(nb / 2) / the rest;
//           c_0 <- (float)log_f64(max(energy, FLT_EPSILON))
// (the chunks are what lets the 16 lanes of a frame share the sums evenly: a chunk is one lane's chain)
#include <float.h>
#include <stdlib.h>

#include "fb_device.h"
#include "fb_kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#define FB_F32_WAVES 16
#define FB_F32_SLOTS 272  // complex slots per frame buffer: 256 + one pad per 16 (conflict-free 16 x 16 transpose)

#define FB_F32_MEL_CHUNK 12  // weights per mel piece (three 16-byte LDS reads)
#define FB_F32_PIECES 64     // at most: four per lane of a frame -- mel pieces and DCT half rows alike
#define FB_F32_MAX_FILTER_PIECES 6  // piece sums k_mfcc_f32 adds per filter (a filter of <= 72 FFT bins)
struct MfccF32Lds {  // offsets in floats
  int tw, twf, win, melw, pfirst, mp0, mcnt, dctp, lift, wave0;
};
__host__ __device__ inline MfccF32Lds fb_mfcc_f32_layout(int L, int nb, int nc, int /*melw_n*/) {
  MfccF32Lds o;
  int off = 0;
  o.tw = off; off += 2 * 256;           // exp(-2 pi i t k1 / 256) at [16 k1 + t]
  o.twf = off; off += 2 * 257;          // exp(-2 pi i k / 512), k <= 256
  off = (off + 1) & ~1;
  o.win = off; off += (L + 1) & ~1;
  off = (off + 3) & ~3;
  o.melw = off; off += FB_F32_PIECES * FB_F32_MEL_CHUNK;  // piece p: 12 weights of one filter, zero-padded
  o.dctp = off; off += FB_F32_PIECES * 16;                // piece 2 c + h: half row h of coefficient c, zero-padded to 16
  o.pfirst = off; off += FB_F32_PIECES;                   // first power-spectrum bin of piece p
  o.mp0 = off; off += 32;                                 // filter m: its first piece ...
  o.mcnt = off; off += 32;                                // ... and how many it has
  o.lift = off; off += 32;
  (void)nb; (void)nc;
  off = (off + 3) & ~3;
  o.wave0 = off;
  return o;
}
// pieces of FB_F32_MEL_CHUNK weights the mel filters split into (k_mfcc_f32 takes at most FB_F32_PIECES)
// A filter of more than FB_F32_MAX_FILTER_PIECES pieces is reported as FB_F32_PIECES + 1: the kernel adds that many piece
// sums per filter and no more (round-5 advisor finding: the total alone was checked, the fifth and sixth piece dropped).
int fb_mfcc_f32_mel_pieces(const int *mel_len, int nb) {
  int np = 0;
  for (int m = 0; m < nb; ++m) {
    const int cnt = mel_len[m] > 0 ? (mel_len[m] + FB_F32_MEL_CHUNK - 1) / FB_F32_MEL_CHUNK : 1;  // (an empty filter: one zero piece)
    if (cnt > FB_F32_MAX_FILTER_PIECES) return FB_F32_PIECES + 1;
    np += cnt;
  }
  return np;
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
// Packed float32 forms.  hipcc finds v_pk_add_f32 for a plain complex add, but builds the crosswise ones (b -+ i d, the
// complex product, the 8th roots) out of all four half combinations plus moves; the VOP3P modifiers say them directly:
// op_sel / op_sel_hi pick the source half that feeds the low / high result, neg_lo / neg_hi flip a source's sign there.
// Each is the same IEEE operation on the same operands as the scalar text in the header (a sign flip is exact).
__device__ __forceinline__ f32x2 fb_sub_i32(f32x2 b, f32x2 d) {  // b - i d = (b.x + d.y, b.y - d.x)
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(b), "v"(d));
  return r;
}
__device__ __forceinline__ f32x2 fb_add_i32(f32x2 b, f32x2 d) {  // b + i d = (b.x - d.y, b.y + d.x)
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(b), "v"(d));
  return r;
}
__device__ __forceinline__ f32x2 fb_cmul32(f32x2 a, f32x2 w) {  // (fmaf(-a.y, w.y, a.x w.x), fmaf(a.y, w.x, a.x w.y))
  f32x2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// forward 4-point DFT in place
__device__ __forceinline__ void fb_dft4_32(f32x2 &v0, f32x2 &v1, f32x2 &v2, f32x2 &v3) {
  const f32x2 a = v0 + v2, b = v0 - v2, c = v1 + v3, d = v1 - v3;
  v0 = a + c;
  v2 = a - c;
  v1 = fb_sub_i32(b, d);
  v3 = fb_add_i32(b, d);
}
// forward 16-point DFT in registers, natural order in and out: 4 x dft4 over n1 (n = 4 n1 + n2), twiddles W16^(n2 k1),
// 4 x dft4 over n2 (k = k1 + 4 k2)
__device__ __forceinline__ void fb_dft16_32(f32x2 (&v)[16]) {
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) fb_dft4_32(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);  // -> A[n2][k1] at v[4 k1 + n2]
  constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, R2 = 0.70710678118654752440f;
  const f32x2 W1 = {C1, -S1}, W3 = {S1, -C1}, W9 = {-C1, S1};   // W16^m = (cos(pi m / 8), -sin(pi m / 8))
  const f32x2 R2P = {R2, R2}, R2N = {R2, -R2};
  auto mul_w2 = [&](f32x2 a) { return R2P * fb_sub_i32(a, a); };   // (1 - i) / sqrt 2: (R2 (a.x + a.y), R2 (a.y - a.x))
  auto mul_w4 = [&](f32x2 a) { return f32x2{a.y, -a.x}; };         // -i
  auto mul_w6 = [&](f32x2 a) {                                     // -(1 + i) / sqrt 2: (R2 (a.y - a.x), -(R2 (a.x + a.y)))
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(a));
    return R2N * r;
  };
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

#ifdef FB_MFCC_STAMP  // instrumented build (tools/profile/mfcc_instrumented.sh): where a wave of k_mfcc_f32 spends its time
__device__ unsigned long long g_mfcc_stamps[4 * 16 * 12];
extern "C" int fb_debug_mfcc_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mfcc_stamps), sizeof(g_mfcc_stamps)) == hipSuccess ? 0 : -1;
}
#define MF_STAMP(k) do { if ((blockIdx.x & 63) == 0 && lane == 0) g_mfcc_stamps[((blockIdx.x >> 6) * 16 + w) * 12 + (k)] = wall_clock64(); } while (0)
#else
#define MF_STAMP(k) do { } while (0)
#endif

// NFULL: the points a < NFULL lie inside the frame for every lane (32 a + 31 < L), their validity selects are dropped
// at compile time (12 of 16 for the recipe's L = 400); 0 = no assumption.
template <int NFULL>
__global__ __launch_bounds__(64 * FB_F32_WAVES, 1) void k_mfcc_f32(FbFrontendDev fe, int melw_n,
                                                                   const int16_t *__restrict__ wav,
                                                                   const int4 *__restrict__ frame_rec,
                                                                   int total_frames, float *__restrict__ mfcc, int words,
                                                                   int uni_T, int uni_n, long long uni_base) {
  if (fe.stop && *fe.stop) return;
  extern __shared__ __attribute__((aligned(16))) float smem32[];
  constexpr int NT = 64 * FB_F32_WAVES, Nc = 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t_lane = lane & 15, fq = lane >> 4;
  const int L = fe.L, nb = fe.nb, nc = fe.nc;
  MF_STAMP(0);
  const MfccF32Lds lo = fb_mfcc_f32_layout(L, nb, nc, melw_n);
  f32x2 *s_tw = reinterpret_cast<f32x2 *>(smem32 + lo.tw);
  f32x2 *s_twf = reinterpret_cast<f32x2 *>(smem32 + lo.twf);
  float *s_win = smem32 + lo.win;
  const float *s_melw = smem32 + lo.melw;
  const float *s_dctp = smem32 + lo.dctp;
  const float *s_lift = smem32 + lo.lift;
  const int *s_pfirst = reinterpret_cast<const int *>(smem32 + lo.pfirst);
  const int *s_mp0 = reinterpret_cast<const int *>(smem32 + lo.mp0), *s_mcnt = reinterpret_cast<const int *>(smem32 + lo.mcnt);
  f32x2 *X = reinterpret_cast<f32x2 *>(smem32 + lo.wave0) + ((size_t)w * 4 + fq) * FB_F32_SLOTS;  // this frame's buffer
  const int n_groups = (total_frames + 3) >> 2;
  const int w_glob = blockIdx.x * FB_F32_WAVES + w, w_step = gridDim.x * FB_F32_WAVES;
  // samples of points p = 16 a + t of frame 4 g + fq: s0 = 32 a + 2 t and s0 + 1, packed into one register per point
  auto load_group = [&](int g, int tl, int (&xq)[16]) {
    const int fl = 4 * g + fq;
    const int fc = fl < total_frames ? fl : total_frames - 1;
    int64_t abs_start;
    int start, n;
    if (uni_T > 0) {  // equal-length utterances (every NES batch): what prepare_batch wrote into the record, computed
      const int b = fc / uni_T, tt = fc - b * uni_T;
      start = fe.snip_edges ? tt * fe.shift : tt * fe.shift + fe.shift / 2 - L / 2;
      n = uni_n;
      abs_start = uni_base + (int64_t)b * uni_n + start;
    } else {
      const int4 rec = frame_rec[fc];
      abs_start = ((int64_t)(unsigned)rec.x) | ((int64_t)rec.y << 32);
      start = rec.z;
      n = rec.w;
    }
    const bool interior = start >= 0 && start + L <= n;
    if (words && __all(interior && !(abs_start & 1))) {  // the pair of a point is one aligned 32-bit word, already in register format
      const int *fr = reinterpret_cast<const int *>(wav + abs_start);
#pragma unroll
      for (int a = 0; a < 16; ++a) xq[a] = fr[a < NFULL ? 16 * a + tl : min(16 * a + tl, (L >> 1) - 1)];
    } else if (__all(interior)) {
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
  MF_STAMP(1);
  __syncthreads();
  MF_STAMP(2);
  const float pre = (float)fe.preemph;
  const int hl = (nb + 1) >> 1;                 // DCT: m < hl is the first half row
  const int n_dslots = (2 * nc + 15) >> 4;      // slots of 16 pieces
  const int n_mslots = __builtin_amdgcn_readfirstlane(s_mcnt[31]);
  {  // the 16 pad slots of this lane's frame buffer are never written by the transposes: keep them finite (a zero weight
     // may meet whatever lies behind the last power-spectrum bin or log mel energy)
    X[17 * t_lane + 16] = f32x2{0.0f, 0.0f};
  }

  for (int g = w_glob; g < n_groups; g += w_step) {
    int t = t_lane;
    asm volatile("" : "+v"(t));  // (keeps the lane-dependent index arithmetic out of registers across trips)
    const int f = 4 * g + fq;
    const bool fvalid = f < total_frames;
    int xp[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) xp[a] = xn[a];
    if (g + w_step < n_groups) load_group(g + w_step, t, xn);
    // ---- exact integer moments of the frame: sum x (|.| < 2^24) and sum x^2 (< 2^39).  A register holds the sample
    //      pair of a point, so one v_dot2 gives x0 + x1 and one gives x0^2 + x1^2 (<= 2^31: read as unsigned); the squares
    //      are summed in float64, where every partial sum is an integer below 2^53, i.e. exact.  L is even, so the two
    //      samples of a pair lie inside the frame or outside it together.
    int isum = 0;
    double sqd = 0.0;
    const s16x2 ones = {1, 1};
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      if (a >= NFULL) xp[a] = 32 * a + 2 * t < L ? xp[a] : 0;
      const s16x2 q = __builtin_bit_cast(s16x2, xp[a]);
      isum = __builtin_amdgcn_sdot2(q, ones, isum, false);
      sqd += (double)(unsigned)__builtin_amdgcn_sdot2(q, q, 0, false);
    }
    isum = fb_row_sum_i32f(isum);
    sqd += fb_dpp_f64<0xb1, 0xf>(sqd);
    sqd += fb_dpp_f64<0x4e, 0xf>(sqd);
    sqd += fb_dpp_f64<0x141, 0xf>(sqd);
    sqd += fb_dpp_f64<0x140, 0xf>(sqd);
    const double dcd = fe.remove_dc ? (double)isum : 0.0;
    const double energy = ((double)L * sqd - dcd * dcd) / (double)L;  // exact numerator (< 2^53), one rounding
    const float mean = fe.remove_dc ? (float)isum / (float)L : 0.0f;

    f32x2 v[16];
    float prev_rot = 0.0f;  // row-rotated second samples (mean removed) of point a - 1
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int s0 = 32 * a + 2 * t;
      const f32x2 d = f32x2{(float)(int)(short)xp[a], (float)(xp[a] >> 16)} - mean;
      // row_ror:1: lane t gets lane (t - 1) & 15 -- the sample before this lane's first one, already converted
      const float rot = __int_as_float(fb_dpp_i32<0x121, 0xf>(__float_as_int(d.y)));
      const f32x2 pv = {t == 0 ? (a == 0 ? d.x : prev_rot) : rot, d.x};  // Kaldi: sample 0 is pre-emphasised with itself
      prev_rot = rot;
      const bool inside = a < NFULL;  // compile time
      const f32x2 wq = *reinterpret_cast<const f32x2 *>(&s_win[inside ? s0 : min(s0, (L - 1) & ~1)]);  // L even: the pair exists
      const f32x2 y = (d - pre * pv) * wq;
      v[a] = inside || s0 < L ? y : f32x2{0.0f, 0.0f};
    }

    MF_STAMP(3);
    // ---- 256-point FFT = radix-16 over a, twiddle W256^(t k1), transpose, radix-16 over b
    fb_dft16_32(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[k1] = fb_cmul32(v[k1], s_tw[16 * k1 + t]);  // W256^(t k1): one row per k1, no index arithmetic
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) X[fb_f32_phys(16 * k1 + t)] = v[k1];
    fb_wave_sync32();
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = X[fb_f32_phys(16 * t + b)];
    MF_STAMP(4);
    fb_dft16_32(v);  // v[k2] = Z[t + 16 k2]
    MF_STAMP(5);
    fb_wave_sync32();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) X[fb_f32_phys(t + 16 * k2)] = v[k2];
    fb_wave_sync32();
    // ---- real-FFT unpack + power spectrum of bins k = t + 16 k2 (and bin 256 in lane t = 0)
    float pwv[17];
    // mirror bins Z[(256 - k) & 255] of k = t + 16 k2: slot 17 (15 - k2) + (16 - t) for t > 0, 17 (16 - k2) for t = 0
    const int mb = t == 0 ? 17 : 16 - t;
    const f32x2 *Xm = X + mb;
#pragma unroll
    for (int k2 = 0; k2 < 17; ++k2) {
      const int kc = k2 < 16 ? t + 16 * k2 : Nc;
      const f32x2 zk = k2 < 16 ? v[k2 & 15] : X[0];
      const f32x2 zr = k2 == 0 ? X[t == 0 ? 0 : 255 + mb] : (k2 < 16 ? Xm[17 * (15 - (k2 & 15))] : X[0]);
      // X[k] = E + W O with E = (Z[k] + conj Z[N-k]) / 2, O = -i (Z[k] - conj Z[N-k]) / 2; the halvings applied once,
      // as 1/4 of the power (exact).  (er, ei) = (zk.x + zr.x, zk.y - zr.y), (dr, di) = (zk.x - zr.x, zk.y + zr.y),
      // xr = er + fmaf(wk.x, di, wk.y dr), xi = ei + fmaf(wk.y, di, -(wk.x dr)) -- two values per instruction
      const f32x2 wk = s_twf[kc];
      f32x2 e2, d2, tt, uu;
      asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(e2) : "v"(zk), "v"(zr));
      asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d2) : "v"(zk), "v"(zr));
      asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(tt) : "v"(wk), "v"(d2));
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(uu) : "v"(wk), "v"(d2), "v"(tt));
      const f32x2 xx = e2 + uu;
      pwv[k2] = 0.25f * __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);
    }
    MF_STAMP(6);
    fb_wave_sync32();
    float *PW = reinterpret_cast<float *>(X);  // 257 floats (+ 7 zeros), then LM, then the mel piece sums
    float *LM = PW + 264;                      // log mel energies: m < hl at [m], the rest at [16 + m - hl]; log energy at [32]
    float *PS = PW + 304;                      // [16 slot + t]
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) PW[t + 16 * k2] = pwv[k2];
    if (t < 8) PW[Nc + t] = t == 0 ? pwv[16] : 0.0f;
    fb_wave_sync32();
    // ---- mel filterbank: piece p = 16 slot + t is FB_F32_MEL_CHUNK consecutive weights of one filter (zero-padded: a
    //      padded step is fmaf(0, finite, e) = e); the lanes' chains are independent, up to four per lane side by side
    {
      float ch[4];
      typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        ch[sl] = 0.0f;
        if (sl < n_mslots) {  // wave-uniform
          const int pc = 16 * sl + t;
          const f32x4 *wp = reinterpret_cast<const f32x4 *>(s_melw + FB_F32_MEL_CHUNK * pc);
          const f32x4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
          const float *pp = PW + s_pfirst[pc];
          float e = 0.0f;
          e = __builtin_fmaf(w0.x, pp[0], e); e = __builtin_fmaf(w0.y, pp[1], e); e = __builtin_fmaf(w0.z, pp[2], e); e = __builtin_fmaf(w0.w, pp[3], e);
          e = __builtin_fmaf(w1.x, pp[4], e); e = __builtin_fmaf(w1.y, pp[5], e); e = __builtin_fmaf(w1.z, pp[6], e); e = __builtin_fmaf(w1.w, pp[7], e);
          e = __builtin_fmaf(w2.x, pp[8], e); e = __builtin_fmaf(w2.y, pp[9], e); e = __builtin_fmaf(w2.z, pp[10], e); e = __builtin_fmaf(w2.w, pp[11], e);
          ch[sl] = e;
        }
      }
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
        if (sl < n_mslots) PS[16 * sl + t] = ch[sl];
    }
    fb_wave_sync32();
    MF_STAMP(9);
    // ---- a filter's piece sums added left to right, log: lane t takes filters t and t + 16; the free second slot of
    //      lane 15 (nb <= 31) takes the log of the frame energy
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int m = t + 16 * half;
      float e = 0.0f;
      if (m < nb) {
        const int q0 = s_mp0[m], cnt = s_mcnt[m];
        e = PS[q0];  // (cnt >= 1; an empty filter owns one all-zero piece)
        // (every piece of the filter, left to right; FB_F32_MAX_FILTER_PIECES is what fb_set_frontend admits: Kaldi's 23
        //  bins at 16 kHz have filters of 52 bins = 5 pieces, 16 bins 67 = 6)
        const float e1 = PS[q0 + (cnt > 1 ? 1 : 0)], e2 = PS[q0 + (cnt > 2 ? 2 : 0)], e3 = PS[q0 + (cnt > 3 ? 3 : 0)];
        const float e4 = PS[q0 + (cnt > 4 ? 4 : 0)], e5 = PS[q0 + (cnt > 5 ? 5 : 0)];
        static_assert(FB_F32_MAX_FILTER_PIECES == 6, "one term per admitted piece");
        if (cnt > 1) e += e1;
        if (cnt > 2) e += e2;
        if (cnt > 3) e += e3;
        if (cnt > 4) e += e4;
        if (cnt > 5) e += e5;
      }
      const bool is_energy = half == 1 && t == 15;
      double ed = is_energy ? energy : (double)e;
      if (ed < (double)FLT_EPSILON) ed = (double)FLT_EPSILON;
      const double le = fb_log_f64(ed);
      if (m < nb) LM[m < hl ? m : 16 + m - hl] = (float)le;
      if (is_energy) LM[32] = (float)(le < fe.log_energy_floor ? fe.log_energy_floor : le);
    }
    fb_wave_sync32();
    MF_STAMP(7);
    // ---- DCT-II, lifter, C0 <- log energy: piece 16 slot + t = half row (t & 1) of coefficient (16 slot + t) / 2 (the
    //      table row is zero beyond the half's length and beyond nc); a quad_perm adds the two halves of a coefficient
    {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 *lp = reinterpret_cast<const f32x4 *>(LM + 16 * (t & 1));
      const f32x4 l0 = lp[0], l1 = lp[1], l2 = lp[2], l3 = lp[3];
      const float len_ = LM[32];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        if (sl < n_dslots) {  // wave-uniform
          const int pc = 16 * sl + t, c = pc >> 1;
          const f32x4 *dp = reinterpret_cast<const f32x4 *>(s_dctp + 16 * pc);
          const f32x4 d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3];
          float acc = 0.0f;
          acc = __builtin_fmaf(d0.x, l0.x, acc); acc = __builtin_fmaf(d0.y, l0.y, acc); acc = __builtin_fmaf(d0.z, l0.z, acc); acc = __builtin_fmaf(d0.w, l0.w, acc);
          acc = __builtin_fmaf(d1.x, l1.x, acc); acc = __builtin_fmaf(d1.y, l1.y, acc); acc = __builtin_fmaf(d1.z, l1.z, acc); acc = __builtin_fmaf(d1.w, l1.w, acc);
          acc = __builtin_fmaf(d2.x, l2.x, acc); acc = __builtin_fmaf(d2.y, l2.y, acc); acc = __builtin_fmaf(d2.z, l2.z, acc); acc = __builtin_fmaf(d2.w, l2.w, acc);
          acc = __builtin_fmaf(d3.x, l3.x, acc); acc = __builtin_fmaf(d3.y, l3.y, acc); acc = __builtin_fmaf(d3.z, l3.z, acc); acc = __builtin_fmaf(d3.w, l3.w, acc);
          const float other = __int_as_float(fb_dpp_i32<0xb1, 0xf>(__float_as_int(acc)));  // quad_perm [1,0,3,2]
          const float tot = (t & 1) ? other + acc : acc + other;                             // first half + second half
          float o = tot * s_lift[c < 32 ? c : 31];
          if (c == 0 && fe.use_energy) o = len_;
          if (fvalid && !(t & 1) && c < nc) mfcc[(size_t)f * nc + c] = o;
        }
      }
    }
    fb_wave_sync32();
    MF_STAMP(8);
  }
}

std::vector<float> fb_mfcc_f32_table(int L, int nb, int nc, const double *window, const double *tw_half, const double *tw_full,
                                     const int *mel_first, const int *mel_len, const int *mel_off, const double *mel_w, int melw_n,
                                     const double *dct, const double *lifter) {
  const MfccF32Lds lo = fb_mfcc_f32_layout(L, nb, nc, melw_n);
  std::vector<float> t((size_t)lo.wave0 + 4, 0.0f);
  for (int k1 = 0; k1 < 16; ++k1)  // exp(-2 pi i (t k1) / 256) at [16 k1 + t]
    for (int tt = 0; tt < 16; ++tt) {
      const int m = (tt * k1) & 255;
      t[lo.tw + 2 * (16 * k1 + tt)] = (float)tw_half[2 * m];
      t[lo.tw + 2 * (16 * k1 + tt) + 1] = (float)tw_half[2 * m + 1];
    }
  for (int i = 0; i <= 256; ++i) { t[lo.twf + 2 * i] = (float)tw_full[2 * i]; t[lo.twf + 2 * i + 1] = (float)tw_full[2 * i + 1]; }
  for (int i = 0; i < L; ++i) t[lo.win + i] = (float)window[i];
  (void)melw_n;
  {  // mel pieces in filter order
    int *pf = reinterpret_cast<int *>(&t[lo.pfirst]), *mp0 = reinterpret_cast<int *>(&t[lo.mp0]), *mcnt = reinterpret_cast<int *>(&t[lo.mcnt]);
    int pc = 0;
    for (int m = 0; m < nb; ++m) {
      const int cnt = mel_len[m] > 0 ? (mel_len[m] + FB_F32_MEL_CHUNK - 1) / FB_F32_MEL_CHUNK : 1;
      mp0[m] = pc;
      mcnt[m] = cnt;
      for (int j = 0; j < cnt && pc < FB_F32_PIECES; ++j, ++pc) {
        pf[pc] = mel_first[m] + FB_F32_MEL_CHUNK * j;
        for (int i = 0; i < FB_F32_MEL_CHUNK; ++i) {
          const int k = FB_F32_MEL_CHUNK * j + i;
          t[lo.melw + FB_F32_MEL_CHUNK * pc + i] = k < mel_len[m] ? (float)mel_w[mel_off[m] + k] : 0.0f;
        }
      }
    }
    mcnt[31] = (pc + 15) / 16;  // slots of 16 pieces (nb <= 31: the entry is free); pieces behind the last one stay zero, bin 0
  }
  {  // DCT half rows: piece 2 c + h = coefficients m in [0, hl) / [hl, nb) of row c, zero-padded to 16
    const int hl = (nb + 1) / 2;
    for (int c = 0; c < nc; ++c)
      for (int h = 0; h < 2; ++h)
        for (int j = 0; j < 16; ++j) {
          const int m = h * hl + j;
          t[lo.dctp + 16 * (2 * c + h) + j] = (j < (h ? nb - hl : hl)) ? (float)dct[c * nb + m] : 0.0f;
        }
  }
  for (int i = 0; i < nc; ++i) t[lo.lift + i] = (float)lifter[i];
  return t;
}
// true when the configuration is one k_mfcc_f32 takes (the recipe's: P = 512, raw energy, at most 31 mel bins / 32
// cepstra, even frame length); fb_set_frontend refuses mfcc_f32 = 1 otherwise
bool fb_mfcc_f32_supported(const FbFrontendDev &fe) {
  return fe.P == 512 && fe.nb <= 31 && fe.nc <= 32 && (fe.L & 1) == 0 && fe.L >= 2 && fe.L <= 512 && fe.raw_energy != 0 && fe.f32_tab != nullptr;
}
bool fb_launch_mfcc_f32(hipStream_t s, const FbFrontendDev &fe, int melw_n, const int16_t *wav, const int32_t *frame_rec,
                        int total_frames, float *mfcc, int uni_T, int64_t uni_n, int64_t uni_base) {
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
  // compute units the launch may take (fe.mfcc_cus, 0 = all 256): a workgroup fills its unit (16 waves at 122 registers), so on
  // a GPU shared by several attacks the launch is held to half of them and walks its frames in two rounds (fb_engine.hip)
  const int cus = fe.mfcc_cus > 0 && fe.mfcc_cus < 256 ? fe.mfcc_cus : 256;
  const int rounds = (n_groups + cus * FB_F32_WAVES - 1) / (cus * FB_F32_WAVES);
  const int blocks = (n_groups + rounds * FB_F32_WAVES - 1) / (rounds * FB_F32_WAVES);
  const int4 *rec = reinterpret_cast<const int4 *>(frame_rec);
  const int words = getenv("FB_MFCC_HALFWORDS") == nullptr;  // (A/B and the parity test of the 16-bit load path)
  if (uni_n > 0x7fffffffLL || getenv("FB_MFCC_RECORDS") != nullptr) uni_T = 0;  // (A/B: always load the records)
  const int un = (int)uni_n;
  const long long ub = (long long)uni_base;
  if (fe.L / 32 >= 12) hipLaunchKernelGGL((k_mfcc_f32<12>), dim3(blocks), dim3(64 * FB_F32_WAVES), shm, s, fe, melw_n, wav, rec, total_frames, mfcc, words, uni_T, un, ub);
  else hipLaunchKernelGGL((k_mfcc_f32<0>), dim3(blocks), dim3(64 * FB_F32_WAVES), shm, s, fe, melw_n, wav, rec, total_frames, mfcc, words, uni_T, un, ub);
  return true;
}
