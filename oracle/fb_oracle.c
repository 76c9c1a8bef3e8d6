/*
 * fb_oracle.c -- CPU restatement of the FAKEBOB NES hot path (see fb_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked or called by the product.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; contraction is OFF
 * so every a*b+c below is two roundings unless written as fma()).
 *
 * Reference citations are into /root/reference (FAKEBOB-adversarial-attack/
 * FAKEBOB).  "[EXT]" marks arithmetic that the reference delegates to Kaldi
 * executables and that is restated here from Kaldi's published algorithms
 * (SURVEY.md Appendix A): parity for those parts is UNPINNED.
 */
#include "fb_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ cfg */
void fbo_default_cfg(fbo_frontend_cfg *c) {
  /* [EXT] voxceleb/v1 conf/mfcc.conf, conf/vad.conf, delta_opts
   * (SURVEY.md A.1, A.4, A.5); cmn flags: gmm_ubm_kaldiHelper.py:196 */
  c->sample_freq = 16000.0;
  c->frame_length = 400;
  c->frame_shift = 160;
  c->padded_length = 512;
  c->num_mel_bins = 30;
  c->num_ceps = 24;
  c->low_freq = 20.0;
  c->high_freq = 7600.0;
  c->preemph = 0.97;
  c->cepstral_lifter = 22.0;
  c->snip_edges = 0;
  c->remove_dc = 1;
  c->use_energy = 1;
  c->raw_energy = 1;
  c->energy_floor = 0.0;
  c->vad_energy_threshold = 5.5;
  c->vad_energy_mean_scale = 0.5;
  c->vad_proportion_threshold = 0.12;
  c->vad_frames_context = 2;
  c->delta_window = 3;
  c->delta_order = 2;
  c->cmn_window = 300;
  c->text_scores = 0;
  c->compress_feats = 0;
  c->mfcc_f32 = 0;
}

/* --------------------------------------------------------------- Philox */
/* Philox4x32-10 (Salmon et al., SC'11).  The reference never seeds its RNG
 * (FAKEBOB.py:234 uses the global numpy state), so "same inputs" must include
 * the noise: this counter-based stream is the shared contract.             */
void fbo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Box-Muller in float32 built ONLY from exactly-rounded primitives
 * (int->float, fmaf, *, +, sqrtf) so that the HIP twin is bit-identical. */
static float fbo_ln_u(float u) { /* u in (0,1] */
  union { float f; uint32_t i; } v; v.f = u;
  int e = (int)(v.i >> 23) - 127;
  v.i = (v.i & 0x007FFFFFu) | 0x3F800000u;
  float m = v.f;
  if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
  float t = m - 1.0f; /* exact */
  /* ln(1+t) = t - t^2/2 + t^3/3 - ... (Horner, 20 terms) */
  float p = -1.0f / 20.0f;
  p = fmaf(p, t, 1.0f / 19.0f);
  p = fmaf(p, t, -1.0f / 18.0f);
  p = fmaf(p, t, 1.0f / 17.0f);
  p = fmaf(p, t, -1.0f / 16.0f);
  p = fmaf(p, t, 1.0f / 15.0f);
  p = fmaf(p, t, -1.0f / 14.0f);
  p = fmaf(p, t, 1.0f / 13.0f);
  p = fmaf(p, t, -1.0f / 12.0f);
  p = fmaf(p, t, 1.0f / 11.0f);
  p = fmaf(p, t, -1.0f / 10.0f);
  p = fmaf(p, t, 1.0f / 9.0f);
  p = fmaf(p, t, -1.0f / 8.0f);
  p = fmaf(p, t, 1.0f / 7.0f);
  p = fmaf(p, t, -1.0f / 6.0f);
  p = fmaf(p, t, 1.0f / 5.0f);
  p = fmaf(p, t, -1.0f / 4.0f);
  p = fmaf(p, t, 1.0f / 3.0f);
  p = fmaf(p, t, -1.0f / 2.0f);
  p = fmaf(p, t, 1.0f);
  p = p * t;
  return fmaf((float)e, 0.693147182f, p);
}

static void fbo_sincos_q(float a, float *s, float *c) { /* a in [0, pi/2) */
  float a2 = a * a;
  float ps = -1.0f / 1307674368000.0f;           /* -1/15! */
  ps = fmaf(ps, a2, 1.0f / 6227020800.0f);       /*  1/13! */
  ps = fmaf(ps, a2, -1.0f / 39916800.0f);        /* -1/11! */
  ps = fmaf(ps, a2, 1.0f / 362880.0f);           /*  1/9!  */
  ps = fmaf(ps, a2, -1.0f / 5040.0f);            /* -1/7!  */
  ps = fmaf(ps, a2, 1.0f / 120.0f);
  ps = fmaf(ps, a2, -1.0f / 6.0f);
  ps = fmaf(ps, a2, 1.0f);
  *s = ps * a;
  float pc = 1.0f / 20922789888000.0f;           /*  1/16! */
  pc = fmaf(pc, a2, -1.0f / 87178291200.0f);     /* -1/14! */
  pc = fmaf(pc, a2, 1.0f / 479001600.0f);        /*  1/12! */
  pc = fmaf(pc, a2, -1.0f / 3628800.0f);         /* -1/10! */
  pc = fmaf(pc, a2, 1.0f / 40320.0f);            /*  1/8!  */
  pc = fmaf(pc, a2, -1.0f / 720.0f);
  pc = fmaf(pc, a2, 1.0f / 24.0f);
  pc = fmaf(pc, a2, -0.5f);
  pc = fmaf(pc, a2, 1.0f);
  *c = pc;
}

static void fbo_box_muller(uint32_t r0, uint32_t r1, float *z0, float *z1) {
  float u1 = (float)((r0 >> 8) + 1u) * 5.9604644775390625e-08f; /* 2^-24, (0,1] */
  uint32_t q = r1 >> 30;
  uint32_t fr = (r1 & 0x3FFFFFFFu) >> 6; /* 24 bits */
  float a = (float)fr * 9.36227702e-08f;  /* (pi/2) * 2^-24 */
  float s, c;
  fbo_sincos_q(a, &s, &c);
  float rr = sqrtf(-2.0f * fbo_ln_u(u1));
  float cs, sn;
  switch (q) {
    case 0: cs = c; sn = s; break;
    case 1: cs = -s; sn = c; break;
    case 2: cs = -c; sn = -s; break;
    default: cs = s; sn = -c; break;
  }
  *z0 = rr * cs;
  *z1 = rr * sn;
}

void fbo_noise(uint64_t seed, uint32_t iter, uint32_t stream, int64_t N, int half, float *z) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int j = 0; j < half; ++j) {
    for (int64_t n4 = 0; n4 < (N + 3) / 4; ++n4) {
      uint32_t ctr[4] = {(uint32_t)n4, (uint32_t)j, iter, stream};
      uint32_t r[4];
      float v[4];
      fbo_philox4x32_10(ctr, key, r);
      fbo_box_muller(r[0], r[1], &v[0], &v[1]);
      fbo_box_muller(r[2], r[3], &v[2], &v[3]);
      for (int k = 0; k < 4; ++k) {
        int64_t n = n4 * 4 + k;
        if (n < N) z[(int64_t)j * N + n] = v[k];
      }
    }
  }
}

/* ----------------------------------------------------------- quantise K1 */
/* (audio * 2**(bits-1)).astype(np.int16): gmm_ubm_OSI.py:83-85,
 * ivector_PLDA_OSI.py:113-115.  numpy's C cast on x86-64 truncates toward
 * zero and keeps the low 16 bits (1.0 -> -32768, golden G5). */
void fbo_quantize(const double *x, int64_t n, int bits_per_sample, int16_t *q) {
  double scale = ldexp(1.0, bits_per_sample - 1);
  for (int64_t i = 0; i < n; ++i) {
    double v = x[i] * scale;
    int64_t t;
    if (!(v > -9.2e18 && v < 9.2e18)) t = 0; /* NaN / out of int64: indefinite -> low bits 0 */
    else t = (int64_t)v;                     /* trunc toward zero */
    q[i] = (int16_t)(uint16_t)((uint64_t)t & 0xFFFFu);
  }
}

/* ------------------------------------------------------------- MFCC [EXT] */
int fbo_num_frames(const fbo_frontend_cfg *c, int64_t n) {
  if (c->snip_edges) {
    if (n < c->frame_length) return 0;
    return (int)(1 + (n - c->frame_length) / c->frame_shift);
  }
  return (int)((n + c->frame_shift / 2) / c->frame_shift);
}
int fbo_feat_dim(const fbo_frontend_cfg *c) { return c->num_ceps * (c->delta_order + 1); }

typedef struct {
  int L, P, nb, nc;
  float *window;        /* [L] povey, stored float like Kaldi */
  double *tw_re, *tw_im;/* [P/2] twiddles */
  int *rev;             /* [P] bit reversal */
  int *mel_first, *mel_len; /* [nb] */
  float *mel_w;         /* [nb][P/2] */
  float *dct;           /* [nc][nb] */
  float *lifter;        /* [nc] */
} fbo_mfcc_tables;

static double mel_scale(double f) { return 1127.0 * log(1.0 + f / 700.0); }

static fbo_mfcc_tables *mfcc_tables_new(const fbo_frontend_cfg *c) {
  fbo_mfcc_tables *t = (fbo_mfcc_tables *)calloc(1, sizeof(*t));
  int L = c->frame_length, P = c->padded_length, nb = c->num_mel_bins, nc = c->num_ceps;
  t->L = L; t->P = P; t->nb = nb; t->nc = nc;
  t->window = (float *)malloc(sizeof(float) * L);
  double a = 2.0 * M_PI / (L - 1);
  for (int i = 0; i < L; ++i) t->window[i] = (float)pow(0.5 - 0.5 * cos(a * i), 0.85);
  t->tw_re = (double *)malloc(sizeof(double) * P / 2);
  t->tw_im = (double *)malloc(sizeof(double) * P / 2);
  for (int k = 0; k < P / 2; ++k) {
    t->tw_re[k] = cos(-2.0 * M_PI * k / P);
    t->tw_im[k] = sin(-2.0 * M_PI * k / P);
  }
  t->rev = (int *)malloc(sizeof(int) * P);
  int bits = 0; while ((1 << bits) < P) ++bits;
  for (int i = 0; i < P; ++i) {
    int r = 0;
    for (int b = 0; b < bits; ++b) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
    t->rev[i] = r;
  }
  /* mel banks: triangles linear in mel, evaluated at FFT-bin centres; bins
   * 0..P/2-1 only */
  int nfft = P / 2;
  t->mel_first = (int *)malloc(sizeof(int) * nb);
  t->mel_len = (int *)malloc(sizeof(int) * nb);
  t->mel_w = (float *)calloc((size_t)nb * nfft, sizeof(float));
  double nyq = 0.5 * c->sample_freq;
  double hi = c->high_freq > 0.0 ? c->high_freq : nyq + c->high_freq;
  double bw = c->sample_freq / P;
  double mlo = mel_scale(c->low_freq), mhi = mel_scale(hi);
  double md = (mhi - mlo) / (nb + 1);
  for (int b = 0; b < nb; ++b) {
    double left = mlo + b * md, center = mlo + (b + 1) * md, right = mlo + (b + 2) * md;
    int first = -1, last = -1;
    for (int i = 0; i < nfft; ++i) {
      double mel = mel_scale(bw * i);
      if (mel > left && mel < right) {
        double w = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
        t->mel_w[(size_t)b * nfft + i] = (float)w;
        if (first < 0) first = i;
        last = i;
      }
    }
    t->mel_first[b] = first < 0 ? 0 : first;
    t->mel_len[b] = first < 0 ? 0 : last - first + 1;
  }
  t->dct = (float *)malloc(sizeof(float) * nc * nb);
  for (int k = 0; k < nc; ++k)
    for (int n = 0; n < nb; ++n)
      t->dct[k * nb + n] = (k == 0) ? (float)sqrt(1.0 / nb)
                                    : (float)(sqrt(2.0 / nb) * cos(M_PI / nb * (n + 0.5) * k));
  t->lifter = (float *)malloc(sizeof(float) * nc);
  for (int i = 0; i < nc; ++i)
    t->lifter[i] = c->cepstral_lifter != 0.0
                       ? (float)(1.0 + 0.5 * c->cepstral_lifter * sin(M_PI * i / c->cepstral_lifter))
                       : 1.0f;
  return t;
}
static void mfcc_tables_free(fbo_mfcc_tables *t) {
  free(t->window); free(t->tw_re); free(t->tw_im); free(t->rev);
  free(t->mel_first); free(t->mel_len); free(t->mel_w); free(t->dct); free(t->lifter); free(t);
}

/* plain iterative radix-2 DIT complex FFT, float64 */
static void fft_c2c(const fbo_mfcc_tables *t, double *re, double *im) {
  int P = t->P;
  for (int i = 0; i < P; ++i) {
    int r = t->rev[i];
    if (r > i) { double a = re[i]; re[i] = re[r]; re[r] = a; a = im[i]; im[i] = im[r]; im[r] = a; }
  }
  for (int len = 2; len <= P; len <<= 1) {
    int half = len >> 1, step = P / len;
    for (int s = 0; s < P; s += len)
      for (int k = 0; k < half; ++k) {
        double wr = t->tw_re[k * step], wi = t->tw_im[k * step];
        double xr = re[s + k + half], xi = im[s + k + half];
        double tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
        re[s + k + half] = re[s + k] - tr; im[s + k + half] = im[s + k] - ti;
        re[s + k] += tr; im[s + k] += ti;
      }
  }
}

/* ---- float32 twin (cfg.mfcc_f32): Kaldi's BaseFloat front-end is float32 end to end.
 * The SAME operations in the SAME order as the product's k_mfcc_f32 (fakebob_amd/csrc/frontend_f32_kernels.hip states the
 * order in its header), every one a single IEEE float32 rounding (this file is compiled with -ffp-contract=off; fused
 * operations are explicit fmaf), so the MFCC matrix is bit-comparable:
 *   - frame energy from the exact integer moments: (L sum x^2 - (sum x)^2) / L, one float64 rounding -> C0 (what the
 *     VAD votes on) does not depend on any float32 summation order;
 *   - mean = (float)sum / (float)L; d(s) = x[s] - mean; y[s] = (d(s) - pre d(s-1)) win[s], d(-1) := d(0);
 *   - z[p] = (y[2p], y[2p+1]); 256-point complex DFT as 16 x 16: dft16 over a of z[16 a + t], times W256^(t k1),
 *     dft16 over t; a dft16 is 4 x dft4, the W16 twiddles, 4 x dft4;
 *   - real-FFT unpack, power, mel sums (chunks of 12 weights), log (the polynomial log of the product, float64, then
 *     one rounding to float32), DCT sums (two half rows), lifter.
 * Only what the recipe uses is taken (padded_length 512, raw energy); anything else returns 0 frames. */
typedef struct { float x, y; } c32;
static c32 c32_add(c32 a, c32 b) { c32 r = {a.x + b.x, a.y + b.y}; return r; }
static c32 c32_sub(c32 a, c32 b) { c32 r = {a.x - b.x, a.y - b.y}; return r; }
static c32 c32_mul(c32 a, c32 w) {
  c32 r;
  r.x = fmaf(-a.y, w.y, a.x * w.x);
  r.y = fmaf(a.y, w.x, a.x * w.y);
  return r;
}
static void dft4_32(c32 *v0, c32 *v1, c32 *v2, c32 *v3) {
  c32 a = c32_add(*v0, *v2), b = c32_sub(*v0, *v2), c = c32_add(*v1, *v3), d = c32_sub(*v1, *v3);
  *v0 = c32_add(a, c);
  *v2 = c32_sub(a, c);
  v1->x = b.x + d.y; v1->y = b.y - d.x; /* b - i d */
  v3->x = b.x - d.y; v3->y = b.y + d.x; /* b + i d */
}
static c32 mul_w2_32(c32 a) { const float R2 = 0.70710678118654752440f; c32 r = {R2 * (a.x + a.y), R2 * (a.y - a.x)}; return r; }
static c32 mul_w4_32(c32 a) { c32 r = {a.y, -a.x}; return r; }
static c32 mul_w6_32(c32 a) { const float R2 = 0.70710678118654752440f; c32 r = {R2 * (a.y - a.x), -(R2 * (a.x + a.y))}; return r; }
static void dft16_32(c32 *v) {
  const float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;
  const c32 W1 = {C1, -S1}, W3 = {S1, -C1}, W9 = {-C1, S1};
  for (int n2 = 0; n2 < 4; ++n2) dft4_32(&v[n2], &v[4 + n2], &v[8 + n2], &v[12 + n2]);
  v[5] = c32_mul(v[5], W1);  v[6] = mul_w2_32(v[6]);   v[7] = c32_mul(v[7], W3);
  v[9] = mul_w2_32(v[9]);    v[10] = mul_w4_32(v[10]); v[11] = mul_w6_32(v[11]);
  v[13] = c32_mul(v[13], W3); v[14] = mul_w6_32(v[14]); v[15] = c32_mul(v[15], W9);
  for (int k1 = 0; k1 < 4; ++k1) dft4_32(&v[4 * k1], &v[4 * k1 + 1], &v[4 * k1 + 2], &v[4 * k1 + 3]);
  for (int i = 0; i < 4; ++i)
    for (int j = i + 1; j < 4; ++j) { c32 tmp = v[4 * i + j]; v[4 * i + j] = v[4 * j + i]; v[4 * j + i] = tmp; }
}
/* the product's fb_log_f64 (fakebob_amd/csrc/fb_device.h), operation for operation: x = m 2^e, m in [sqrt(1/2), sqrt 2),
 * log m = 2 atanh((m-1)/(m+1)) as a ten-term odd series, e ln 2 added as hi + lo */
static double log_poly_f64(double x) {
  int e;
  double m = frexp(x, &e);
  if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
  const double sr = (m - 1.0) / (m + 1.0);
  const double z = sr * sr;
  double p = 1.0 / 21.0;
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  const double two_s = sr + sr;
  const double lm = fma(two_s * z, p, two_s);
  const double ed = (double)e;
  return fma(ed, 0x1.62e42fee00000p-1, fma(ed, 0x1.a39ef35793c76p-33, lm));
}

static int fbo_mfcc_f32(const fbo_frontend_cfg *c, const int16_t *wav, int64_t n, float *out) {
  int T = fbo_num_frames(c, n);
  if (T <= 0) return 0;
  if (c->padded_length != 512 || !c->raw_energy || (c->frame_length & 1) || c->frame_length > 512) return 0;
  fbo_mfcc_tables *t = mfcc_tables_new(c);
  const int L = t->L, nb = t->nb, nc = t->nc, Nc = 256;
  c32 *tw = (c32 *)malloc(sizeof(c32) * Nc), *twf = (c32 *)malloc(sizeof(c32) * (Nc + 1));
  for (int m = 0; m < Nc; ++m) { tw[m].x = (float)cos(-2.0 * M_PI * m / Nc); tw[m].y = (float)sin(-2.0 * M_PI * m / Nc); }
  for (int k = 0; k <= Nc; ++k) { twf[k].x = (float)cos(-2.0 * M_PI * k / 512); twf[k].y = (float)sin(-2.0 * M_PI * k / 512); }
  const float pre = (float)c->preemph;
  int32_t x[512];
  float y[512], pw[257], lm[64];
  c32 X[256], Z[256], v[16];
  for (int f = 0; f < T; ++f) {
    int64_t start = c->snip_edges ? (int64_t)f * c->frame_shift
                                  : (int64_t)f * c->frame_shift + c->frame_shift / 2 - L / 2;
    int64_t isum = 0, sumsq = 0;
    for (int s = 0; s < L; ++s) {
      int64_t k = start + s;
      while (k < 0 || k >= n) { if (k < 0) k = -k - 1; else k = 2 * n - 1 - k; }
      x[s] = wav[k];
      isum += x[s];
      sumsq += (int64_t)x[s] * x[s];
    }
    const int64_t dc = c->remove_dc ? isum : 0;
    const double energy = (double)((int64_t)L * sumsq - dc * dc) / (double)L;
    const float mean = c->remove_dc ? (float)isum / (float)L : 0.0f;
    for (int s = 0; s < L; ++s) {
      const float d = (float)x[s] - mean, dp = (float)x[s > 0 ? s - 1 : 0] - mean;
      y[s] = (d - pre * dp) * t->window[s];
    }
    for (int s = L; s < 512; ++s) y[s] = 0.0f;
    /* 16 x 16: first pass over a for every t, twiddle, second pass over t for every k1 */
    for (int tt = 0; tt < 16; ++tt) {
      for (int a = 0; a < 16; ++a) { v[a].x = y[2 * (16 * a + tt)]; v[a].y = y[2 * (16 * a + tt) + 1]; }
      dft16_32(v);
      for (int k1 = 1; k1 < 16; ++k1) v[k1] = c32_mul(v[k1], tw[(tt * k1) & (Nc - 1)]);
      for (int k1 = 0; k1 < 16; ++k1) X[16 * k1 + tt] = v[k1];
    }
    for (int k1 = 0; k1 < 16; ++k1) {
      for (int b = 0; b < 16; ++b) v[b] = X[16 * k1 + b];
      dft16_32(v);
      for (int k2 = 0; k2 < 16; ++k2) Z[k1 + 16 * k2] = v[k2];
    }
    for (int k = 0; k <= Nc; ++k) {
      const c32 zk = Z[k & (Nc - 1)], zr = Z[(Nc - k) & (Nc - 1)], wk = twf[k];
      const float er = zk.x + zr.x, ei = zk.y - zr.y;
      const float dr = zk.x - zr.x, di = zk.y + zr.y;
      const float xr = er + fmaf(wk.x, di, wk.y * dr);
      const float xi = ei + fmaf(wk.y, di, -(wk.x * dr));
      pw[k] = 0.25f * fmaf(xr, xr, xi * xi);
    }
    for (int b = 0; b < nb; ++b) {
      /* the filter's weights in chunks of 12: every chunk an in-order fmaf chain from zero, the chunk sums added left
       * to right (the product gives a chunk to one lane) */
      float e = 0.0f;
      const float *w = t->mel_w + (size_t)b * Nc;
      const int first = t->mel_first[b], len = t->mel_len[b];
      for (int c0 = 0; c0 < len; c0 += 12) {
        float ch = 0.0f;
        for (int i = first + c0; i < first + (c0 + 12 < len ? c0 + 12 : len); ++i) ch = fmaf(w[i], pw[i], ch);
        e = c0 == 0 ? ch : e + ch;
      }
      double ed = (double)e;
      if (ed < (double)FLT_EPSILON) ed = (double)FLT_EPSILON;
      lm[b] = (float)log_poly_f64(ed);
    }
    for (int k = 0; k < nc; ++k) { /* the row as two halves, each an in-order fmaf chain, added */
      const int hl = (nb + 1) / 2;
      float a0 = 0.0f, a1 = 0.0f;
      for (int b = 0; b < hl; ++b) a0 = fmaf(t->dct[k * nb + b], lm[b], a0);
      for (int b = hl; b < nb; ++b) a1 = fmaf(t->dct[k * nb + b], lm[b], a1);
      out[(size_t)f * nc + k] = (a0 + a1) * t->lifter[k];
    }
    if (c->use_energy) {
      double le = log_poly_f64(energy > (double)FLT_EPSILON ? energy : (double)FLT_EPSILON);
      if (c->energy_floor > 0.0 && le < log(c->energy_floor)) le = log(c->energy_floor);
      out[(size_t)f * nc] = (float)le;
    }
  }
  free(tw); free(twf);
  mfcc_tables_free(t);
  return T;
}

int fbo_mfcc(const fbo_frontend_cfg *c, const int16_t *wav, int64_t n, float *out) {
  if (c->mfcc_f32) return fbo_mfcc_f32(c, wav, n, out);
  int T = fbo_num_frames(c, n);
  if (T <= 0) return 0;
  fbo_mfcc_tables *t = mfcc_tables_new(c);
  int L = t->L, P = t->P, nb = t->nb, nc = t->nc, nfft = P / 2;
  double *re = (double *)malloc(sizeof(double) * P), *im = (double *)malloc(sizeof(double) * P);
  double *pw = (double *)malloc(sizeof(double) * (nfft + 1));
  double *lm = (double *)malloc(sizeof(double) * nb);
  for (int f = 0; f < T; ++f) {
    int64_t start = c->snip_edges ? (int64_t)f * c->frame_shift
                                  : (int64_t)f * c->frame_shift + c->frame_shift / 2 - L / 2;
    for (int s = 0; s < L; ++s) {
      int64_t k = start + s;
      while (k < 0 || k >= n) { if (k < 0) k = -k - 1; else k = 2 * n - 1 - k; }
      re[s] = (double)wav[k];
    }
    if (c->remove_dc) {
      double sum = 0.0;
      for (int s = 0; s < L; ++s) sum += re[s];
      double mean = sum / L;
      for (int s = 0; s < L; ++s) re[s] -= mean;
    }
    double energy = 0.0;
    if (c->raw_energy) { for (int s = 0; s < L; ++s) energy += re[s] * re[s]; }
    if (c->preemph != 0.0) {
      for (int s = L - 1; s > 0; --s) re[s] -= c->preemph * re[s - 1];
      re[0] -= c->preemph * re[0];
    }
    for (int s = 0; s < L; ++s) re[s] *= (double)t->window[s];
    if (!c->raw_energy) { for (int s = 0; s < L; ++s) energy += re[s] * re[s]; }
    double log_energy = log(energy > (double)FLT_EPSILON ? energy : (double)FLT_EPSILON);
    for (int s = L; s < P; ++s) re[s] = 0.0;
    for (int s = 0; s < P; ++s) im[s] = 0.0;
    fft_c2c(t, re, im);
    for (int k = 0; k <= nfft; ++k) pw[k] = re[k] * re[k] + im[k] * im[k];
    for (int b = 0; b < nb; ++b) {
      double e = 0.0;
      const float *w = t->mel_w + (size_t)b * nfft;
      for (int i = t->mel_first[b]; i < t->mel_first[b] + t->mel_len[b]; ++i) e += (double)w[i] * pw[i];
      if (e < (double)FLT_EPSILON) e = (double)FLT_EPSILON;
      lm[b] = log(e);
    }
    for (int k = 0; k < nc; ++k) {
      double acc = 0.0;
      for (int b = 0; b < nb; ++b) acc += (double)t->dct[k * nb + b] * lm[b];
      acc *= (double)t->lifter[k];
      out[(size_t)f * nc + k] = (float)acc;
    }
    if (c->use_energy) {
      if (c->energy_floor > 0.0 && log_energy < log(c->energy_floor)) log_energy = log(c->energy_floor);
      out[(size_t)f * nc] = (float)log_energy;
    }
  }
  free(re); free(im); free(pw); free(lm);
  mfcc_tables_free(t);
  return T;
}

/* --------------------------------------------------------------- VAD [EXT] */
/* ---- Kaldi CompressedMatrix (matrix/compressed-matrix.{h,cc}, not part of the reference tree [EXT]).
 * steps/make_mfcc.sh, which the reference calls with its defaults (gmm_ubm_kaldiHelper.py:138-140,
 * ivector_PLDA_kaldiHelper.py:163-165), pipes compute-mfcc-feats into `copy-feats --compress=true`: the archive
 * every later stage reads holds the lossy form.  Restated from the published algorithm:
 *   global header: min, range = max - min of the whole matrix (max = min + 1 + |min| if they coincide)
 *   T > 8  (kSpeechFeature): per column the order statistics at 0, T/4, 3(T/4), T-1 as uint16 fractions of the
 *          global range (forced strictly increasing), every element as one byte: 0..64 | 64..192 | 192..255
 *          linearly between consecutive anchors
 *   T <= 8 (kTwoByteAuto): every element as a uint16 fraction of the global range
 * Float/double promotion follows the C++ expressions literally (float operands, double literals). */
static int fbo_cm_to_u16(float minv, float range, float value) {
  float f = (value - minv) / range;
  if (f > 1.0f) f = 1.0f;
  if (f < 0.0f) f = 0.0f;
  return (int)((double)(f * 65535.0f) + 0.499);
}
static float fbo_cm_from_u16(float minv, float range, int v) {
  return minv + range * 1.52590218966964e-05F * (float)v;
}
static int fbo_cm_to_char(float p0, float p25, float p75, float p100, float value) {
  int ans;
  if (value < p25) {
    float f = (value - p0) / (p25 - p0);
    ans = (int)((double)(f * 64.0f) + 0.5);
    if (ans < 0) ans = 0;
    if (ans > 64) ans = 64;
  } else if (value < p75) {
    float f = (value - p25) / (p75 - p25);
    ans = 64 + (int)((double)(f * 128.0f) + 0.5);
    if (ans < 64) ans = 64;
    if (ans > 192) ans = 192;
  } else {
    float f = (value - p75) / (p100 - p75);
    ans = 192 + (int)((double)(f * 63.0f) + 0.5);
    if (ans < 192) ans = 192;
    if (ans > 255) ans = 255;
  }
  return ans;
}
static float fbo_cm_from_char(float p0, float p25, float p75, float p100, int v) {
  if (v <= 64) return (float)((double)p0 + (double)((p25 - p0) * (float)v) * (1 / 64.0));
  if (v <= 192) return (float)((double)p25 + (double)((p75 - p25) * (float)(v - 64)) * (1 / 128.0));
  return (float)((double)p75 + (double)((p100 - p75) * (float)(v - 192)) * (1 / 63.0));
}
static int fbo_cmp_float(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}
void fbo_compress_roundtrip(float *m, int T, int nc) {
  if (T <= 0 || nc <= 0) return;
  float minv = m[0], maxv = m[0];
  for (size_t i = 0; i < (size_t)T * nc; ++i) {
    if (m[i] < minv) minv = m[i];
    if (m[i] > maxv) maxv = m[i];
  }
  if (maxv == minv) maxv = minv + (1.0f + fabsf(minv));
  const float range = maxv - minv;
  if (T <= 8) {
    for (size_t i = 0; i < (size_t)T * nc; ++i) m[i] = fbo_cm_from_u16(minv, range, fbo_cm_to_u16(minv, range, m[i]));
    return;
  }
  float *col = (float *)malloc(sizeof(float) * (size_t)T);
  const int q = T / 4;
  for (int c = 0; c < nc; ++c) {
    for (int t = 0; t < T; ++t) col[t] = m[(size_t)t * nc + c];
    qsort(col, (size_t)T, sizeof(float), fbo_cmp_float);  /* nth_element leaves the sorted-order values there */
    int u0 = fbo_cm_to_u16(minv, range, col[0]);
    if (u0 > 65532) u0 = 65532;
    int u25 = fbo_cm_to_u16(minv, range, col[q]);
    if (u25 < u0 + 1) u25 = u0 + 1;
    if (u25 > 65533) u25 = 65533;
    int u75 = fbo_cm_to_u16(minv, range, col[3 * q]);
    if (u75 < u25 + 1) u75 = u25 + 1;
    if (u75 > 65534) u75 = 65534;
    int u100 = fbo_cm_to_u16(minv, range, col[T - 1]);
    if (u100 < u75 + 1) u100 = u75 + 1;
    const float p0 = fbo_cm_from_u16(minv, range, u0), p25 = fbo_cm_from_u16(minv, range, u25),
                p75 = fbo_cm_from_u16(minv, range, u75), p100 = fbo_cm_from_u16(minv, range, u100);
    for (int t = 0; t < T; ++t) {
      float *x = &m[(size_t)t * nc + c];
      *x = fbo_cm_from_char(p0, p25, p75, p100, fbo_cm_to_char(p0, p25, p75, p100, *x));
    }
  }
  free(col);
}

void fbo_vad(const fbo_frontend_cfg *c, const float *mfcc, int T, uint8_t *voiced) {
  int nc = c->num_ceps;
  double sum = 0.0;
  for (int t = 0; t < T; ++t) sum += (double)mfcc[(size_t)t * nc];
  float thr = (float)(c->vad_energy_threshold + c->vad_energy_mean_scale * sum / T);
  int ctx = c->vad_frames_context;
  for (int t = 0; t < T; ++t) {
    int num = 0, den = 0;
    for (int t2 = t - ctx; t2 <= t + ctx; ++t2)
      if (t2 >= 0 && t2 < T) { ++den; if (mfcc[(size_t)t2 * nc] > thr) ++num; }
    voiced[t] = ((float)num >= (float)den * (float)c->vad_proportion_threshold) ? 1 : 0;
  }
}

/* ------------------------------------------------------- add-deltas [EXT] */
void fbo_deltas(const fbo_frontend_cfg *c, const float *mfcc, int T, float *out) {
  int nc = c->num_ceps, order = c->delta_order, W = c->delta_window;
  int dim = nc * (order + 1);
  /* scales[i]: kernel of order i, length 2*i*W+1 */
  int maxlen = 2 * order * W + 1;
  double *scales = (double *)calloc((size_t)(order + 1) * maxlen, sizeof(double));
  scales[0] = 1.0;
  for (int i = 1; i <= order; ++i) {
    const double *prev = scales + (size_t)(i - 1) * maxlen;
    double *cur = scales + (size_t)i * maxlen;
    int prev_off = (i - 1) * W, cur_off = i * W;
    double normalizer = 0.0;
    for (int j = -W; j <= W; ++j) {
      normalizer += (double)j * j;
      for (int k = -prev_off; k <= prev_off; ++k)
        cur[j + k + cur_off] += (double)j * prev[k + prev_off];
    }
    for (int k = 0; k < 2 * cur_off + 1; ++k) cur[k] = (double)(float)(cur[k] / normalizer);
  }
  for (int t = 0; t < T; ++t) {
    for (int i = 0; i <= order; ++i) {
      const double *sc = scales + (size_t)i * maxlen;
      int off = i * W;
      for (int d = 0; d < nc; ++d) {
        double acc = 0.0;
        for (int j = -off; j <= off; ++j) {
          int tt = t + j; if (tt < 0) tt = 0; if (tt > T - 1) tt = T - 1;
          double s = sc[j + off];
          if (s != 0.0) acc += s * (double)mfcc[(size_t)tt * nc + d];
        }
        out[(size_t)t * dim + i * nc + d] = (float)acc;
      }
    }
  }
  free(scales);
}

/* ----------------------------------------------- apply-cmvn-sliding [EXT] */
void fbo_cmvn_sliding(const fbo_frontend_cfg *c, float *feats, int T, int dim) {
  int Wn = c->cmn_window;
  float *src = (float *)malloc(sizeof(float) * (size_t)T * dim);
  memcpy(src, feats, sizeof(float) * (size_t)T * dim);
  double *pre = (double *)calloc((size_t)(T + 1) * dim, sizeof(double)); /* prefix sums */
  for (int t = 0; t < T; ++t)
    for (int d = 0; d < dim; ++d)
      pre[(size_t)(t + 1) * dim + d] = pre[(size_t)t * dim + d] + (double)src[(size_t)t * dim + d];
  for (int t = 0; t < T; ++t) {
    int wb = t - Wn / 2, we = wb + Wn;
    if (wb < 0) { we -= wb; wb = 0; }
    if (we > T) { wb -= (we - T); we = T; if (wb < 0) wb = 0; }
    int wf = we - wb;
    float alpha = (float)(-1.0 / wf);
    for (int d = 0; d < dim; ++d) {
      double sum = pre[(size_t)we * dim + d] - pre[(size_t)wb * dim + d];
      feats[(size_t)t * dim + d] = (float)((double)src[(size_t)t * dim + d] + (double)alpha * sum);
    }
  }
  free(src); free(pre);
}

/* pipeline order: gmm_ubm_kaldiHelper.py:195-198 (add-deltas | apply-cmvn-
 * sliding | select-voiced-frames); VAD from the raw MFCC C0 (:151-169). */
int fbo_frontend(const fbo_frontend_cfg *c, const int16_t *wav, int64_t n, float *feats, int *T_out) {
  int T = fbo_num_frames(c, n);
  if (T_out) *T_out = T;
  if (T <= 0) return 0;
  int nc = c->num_ceps, dim = fbo_feat_dim(c);
  float *mf = (float *)malloc(sizeof(float) * (size_t)T * nc);
  float *df = (float *)malloc(sizeof(float) * (size_t)T * dim);
  uint8_t *v = (uint8_t *)malloc(T);
  fbo_mfcc(c, wav, n, mf);
  if (c->compress_feats) fbo_compress_roundtrip(mf, T, nc);
  fbo_vad(c, mf, T, v);
  fbo_deltas(c, mf, T, df);
  fbo_cmvn_sliding(c, df, T, dim);
  int tv = 0;
  for (int t = 0; t < T; ++t)
    if (v[t]) { memcpy(feats + (size_t)tv * dim, df + (size_t)t * dim, sizeof(float) * dim); ++tv; }
  free(mf); free(df); free(v);
  return tv;
}

/* Kaldi text output of a float score (6 significant digits) read back by float(): identical arithmetic to
 * fb_round6 in fakebob_amd/csrc/fb_device.h (table of exact powers of ten + one correctly rounded division) */
double fbo_round6(double xin) {
  const double x = (double)(float)xin;
  if (x == 0.0 || !(x == x) || x - x != 0.0) return x;
  static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const double ax = x < 0.0 ? -x : x;
  int e = 0;
  if (ax >= 1.0) { while (e < 21 && ax >= p10[e + 1]) ++e; }
  else { while (e > -16 && ax * p10[-e] < 1.0) --e; }
  const int k = 5 - e;
  double scaled = k >= 0 ? ax * p10[k > 22 ? 22 : k] : ax / p10[-k];
  double r = rint(scaled);
  int kk = k;
  if (r >= 1e6) { r = rint(r / 10.0); kk -= 1; }
  const double v = kk >= 0 ? r / p10[kk > 22 ? 22 : kk] : r * p10[-kk];
  return x < 0.0 ? -v : v;
}

/* -------------------------------------- gmm-global-get-frame-likes [EXT] */
/* gmm_ubm_kaldiHelper.py:206 runs `gmm-global-get-frame-likes --average=true`: DiagGmm::LogLikelihood per voiced frame.
 * Kaldi (diag-gmm.cc, kaldi-vector.cc) fills a Vector<BaseFloat> `loglikes` -- a float32 storage point --
 * with gconst_k + means_invvars_k . x - 0.5 inv_vars_k . x^2 and returns loglikes.LogSumExp(), which is NOT the plain
 * sum (VectorBase<Real>::LogSumExp with the default prune < 0):
 *     max_elem = Max();  cutoff = max_elem + kMinLogDiffFloat;          kMinLogDiffFloat = Log(FLT_EPSILON) = -15.9424 (float)
 *     double sum = 0;  for every element f >= cutoff:  sum += Exp(f - max_elem);      float difference, float Exp (expf)
 *     return max_elem + Log(sum);                                        Log of the double, result rounded to float
 * i.e. components more than 15.94 nats below the frame's best one are not summed at all.  mode 0 (default) restates
 * exactly that on the float32-rounded component values; mode 1 is the full float64 sum of earlier rounds (what scipy's
 * logsumexp computes: kept for the independent-math checks and to MEASURE the cutoff's effect: <= C 2^-23 relative per
 * frame in the worst case, ~1e-7 on the benchmark models -- tests/test_oracle_frontend.py). */
static int g_lse_full = 0;
void fbo_set_logsumexp(int full_sum) { g_lse_full = full_sum ? 1 : 0; }
int fbo_get_logsumexp(void) { return g_lse_full; }

double fbo_diag_gmm_loglikes(const float *gc, const float *miv, const float *iv, int C, int D,
                             const float *feats, int Tv, float *ll_out) {
  double total = 0.0;
  const int full = g_lse_full;
  const float min_log_diff = logf(FLT_EPSILON);            /* kMinLogDiffFloat */
  double *x = (double *)malloc(sizeof(double) * 2 * D);
  double *ll = (double *)malloc(sizeof(double) * C);
  for (int t = 0; t < Tv; ++t) {
    const float *f = feats + (size_t)t * D;
    for (int d = 0; d < D; ++d) { x[d] = (double)f[d]; x[D + d] = (double)(float)(f[d] * f[d]); }
    double mx = -INFINITY;
    for (int k = 0; k < C; ++k) {
      const float *m = miv + (size_t)k * D, *v = iv + (size_t)k * D;
      double a = 0.0, b = 0.0;
      for (int d = 0; d < D; ++d) { a += (double)m[d] * x[d]; b += (double)v[d] * x[D + d]; }
      double l = (double)gc[k] + a - 0.5 * b;
      if (!full) l = (double)(float)l;                      /* loglikes is a Vector<BaseFloat> */
      ll[k] = l;
      if (l > mx) mx = l;
    }
    float lf;
    if (full) {
      double s = 0.0;
      for (int k = 0; k < C; ++k) s += exp(ll[k] - mx);
      lf = (float)(mx + log(s));
    } else {
      const float max_elem = (float)mx, cutoff = max_elem + min_log_diff;
      double s = 0.0;
      for (int k = 0; k < C; ++k) {
        const float fk = (float)ll[k];
        if (fk >= cutoff) s += (double)expf(fk - max_elem);
      }
      lf = (float)((double)max_elem + log(s));
    }
    if (ll_out) ll_out[t] = lf;
    total += (double)lf;
  }
  free(x); free(ll);
  return total;
}

int fbo_gmm_score_batch(const fbo_frontend_cfg *c, const int16_t *wav, const int64_t *off, int B,
                        const float *gc, const float *miv, const float *iv, int M, int C, int D,
                        double *raw, int *tv_out, int nthreads) {
  int err = 0;
  if (fbo_feat_dim(c) != D) return -1000000;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int b = 0; b < B; ++b) {
    int64_t n = off[b + 1] - off[b];
    int T = fbo_num_frames(c, n);
    float *feats = (float *)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1) * D);
    int Tt = 0;
    int tv = fbo_frontend(c, wav + off[b], n, feats, &Tt);
    if (tv_out) tv_out[b] = tv;
    if (tv <= 0) {
#ifdef _OPENMP
#pragma omp critical
#endif
      { if (err == 0 || -(b + 1) > err) err = -(b + 1); }
      for (int m = 0; m < M; ++m) raw[(size_t)b * M + m] = NAN;
    } else {
      for (int m = 0; m < M; ++m) {
        double tot = fbo_diag_gmm_loglikes(gc + (size_t)m * C, miv + (size_t)m * C * D,
                                           iv + (size_t)m * C * D, C, D, feats, tv, NULL);
        raw[(size_t)b * M + m] = c->text_scores ? fbo_round6(tot / tv) : tot / tv;
      }
    }
    free(feats);
  }
  (void)nthreads;
  return err;
}

/* ------------------- enrolment: gmm-global-acc-stats + MapDiagGmmUpdate [EXT] */
/* build_spk_models.py:184-216.  Per voiced frame: component log-likelihoods (float64 here, Kaldi: float32
 * sgemv), float32 soft-max as Kaldi's ComponentPosteriors / ApplySoftMax (max, sequential sum of exp,
 * scale by 1/sum), posteriors accumulated in float64 (AccumDiagGmm) in frame order. */
int fbo_gmm_acc_stats(const fbo_frontend_cfg *c, const int16_t *wav, int64_t n, const float *gc,
                      const float *miv, const float *iv, int C, int D, double *occ, double *F) {
  int T = fbo_num_frames(c, n);
  if (T <= 0) return -1;
  float *feats = (float *)malloc(sizeof(float) * (size_t)T * D);
  int Tt = 0;
  int tv = fbo_frontend(c, wav, n, feats, &Tt);
  if (tv <= 0) { free(feats); return -1; }
  for (int k = 0; k < C; ++k) occ[k] = 0.0;
  for (size_t i = 0; i < (size_t)C * D; ++i) F[i] = 0.0;
  float *ll = (float *)malloc(sizeof(float) * C);
  for (int t = 0; t < tv; ++t) {
    const float *f = feats + (size_t)t * D;
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) {
      const float *m = miv + (size_t)k * D, *v = iv + (size_t)k * D;
      double a = 0.0, b = 0.0;
      for (int d = 0; d < D; ++d) { a += (double)m[d] * (double)f[d]; b += (double)v[d] * (double)(float)(f[d] * f[d]); }
      ll[k] = (float)((double)gc[k] + a - 0.5 * b);
      if (ll[k] > mx) mx = ll[k];
    }
    float sum = 0.0f;
    for (int k = 0; k < C; ++k) { ll[k] = expf(ll[k] - mx); sum += ll[k]; }
    const float inv = 1.0f / sum;
    for (int k = 0; k < C; ++k) {
      const double p = (double)(float)(ll[k] * inv);
      occ[k] += p;
      double *Fk = F + (size_t)k * D;
      for (int d = 0; d < D; ++d) Fk[d] += p * (double)f[d];
    }
  }
  free(ll); free(feats);
  return tv;
}
/* MapDiagGmmUpdate, update-flags = "m", mean_tau (gmm-global-est-map.cc:31,62-92): only components with
 * occupancy > 0 move: mean' = F/(occ+tau) + tau/(occ+tau) * mean, in float64 (DiagGmmNormal). */
void fbo_map_update_means(const double *means, const double *occ, const double *F, int C, int D, double tau,
                          double *new_means) {
  for (int k = 0; k < C; ++k) {
    for (int d = 0; d < D; ++d) {
      const double old = means[(size_t)k * D + d];
      double v = old;
      if (occ[k] > 0.0) v = F[(size_t)k * D + d] * (1.0 / (occ[k] + tau)) + (tau / (occ[k] + tau)) * old;
      new_means[(size_t)k * D + d] = v;
    }
  }
}

/* --------------------------------------------------------------- NES core */
double fbo_np_sum(const double *a, int64_t n) {
  /* numpy pairwise_sum (loops_utils.h.src), contiguous float64, verified
   * bit-for-bit against numpy 2.2.6 in tests/test_oracle_nes.py */
  if (n < 8) {
    double r = 0.0;
    for (int64_t i = 0; i < n; ++i) r += a[i];
    return r;
  } else if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  } else {
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return fbo_np_sum(a, n2) + fbo_np_sum(a + n2, n - n2);
  }
}

void fbo_loss(int task, int attack_type, const double *score, int B, int S, double threshold,
              double adver_thresh, int target, int true_label, double *loss) {
  for (int b = 0; b < B; ++b) {
    const double *s = score + (size_t)b * S;
    if (task == FBO_TASK_OSI && attack_type == FBO_TARGETED) { /* FAKEBOB.py:256-262 */
      double om = -INFINITY;
      for (int j = 0; j < S; ++j) if (j != target && s[j] > om) om = s[j];
      double mx = om > threshold ? om : threshold; /* np.maximum */
      loss[b] = (mx + adver_thresh) - s[target];
    } else if (task == FBO_TASK_OSI) {                          /* :265-269 */
      double mx = -INFINITY;
      for (int j = 0; j < S; ++j) if (s[j] > mx) mx = s[j];
      loss[b] = (threshold + adver_thresh) - mx;
    } else if (task == FBO_TASK_CSI && attack_type == FBO_TARGETED) { /* :275-281 */
      double om = -INFINITY;
      for (int j = 0; j < S; ++j) if (j != target && s[j] > om) om = s[j];
      loss[b] = (om + adver_thresh) - s[target];
    } else if (task == FBO_TASK_CSI) {                          /* :285-291 */
      double om = -INFINITY;
      for (int j = 0; j < S; ++j) if (j != true_label && s[j] > om) om = s[j];
      loss[b] = (s[true_label] + adver_thresh) - om;
    } else {                                                    /* SV :297 */
      loss[b] = (threshold + adver_thresh) - s[0];
    }
  }
}

int fbo_get_grad(const fbo_nes_params *p, fbo_score_fn fn, void *ctx, const double *audio,
                 int64_t N, const double *noise_pos, uint64_t seed, uint32_t iter,
                 uint32_t stream, double *final_loss, double *grad, double *adver_loss,
                 double *score0) {
  int half = p->samples_per_draw / 2, spd = 2 * half, B = spd + 1, S = p->n_spk;
  float *z = NULL;
  if (!noise_pos) {
    z = (float *)malloc(sizeof(float) * (size_t)N * (half > 0 ? half : 1));
    fbo_noise(seed, iter, stream, N, half, z);
  }
  double *aud = (double *)malloc(sizeof(double) * (size_t)N * B); /* [B][N] */
  double *noise = (double *)malloc(sizeof(double) * (size_t)N * (spd > 0 ? spd : 1)); /* [N][spd] */
  for (int64_t n = 0; n < N; ++n) {
    aud[n] = p->sigma * 0.0 + audio[n];
    for (int j = 0; j < half; ++j) {
      double zz = noise_pos ? noise_pos[(size_t)n * half + j] : (double)z[(size_t)j * N + n];
      double nz = -1.0 * zz;
      noise[(size_t)n * spd + j] = zz;
      noise[(size_t)n * spd + half + j] = nz;
      aud[(size_t)(1 + j) * N + n] = p->sigma * zz + audio[n];        /* FAKEBOB.py:237 */
      aud[(size_t)(1 + half + j) * N + n] = p->sigma * nz + audio[n];
    }
  }
  double *scores = (double *)malloc(sizeof(double) * (size_t)B * S);
  double *loss = (double *)malloc(sizeof(double) * B);
  int rc = fn(ctx, aud, N, B, scores);
  if (rc == 0) {
    fbo_loss(p->task, p->attack_type, scores, B, S, p->threshold, p->adver_thresh, p->target,
             p->true_label, loss);
    *adver_loss = loss[0];
    for (int j = 0; j < S; ++j) score0[j] = scores[j];
    *final_loss = fbo_np_sum(loss + 1, spd) / (double)spd;            /* np.mean :243 */
    double *prod = (double *)malloc(sizeof(double) * (spd > 0 ? spd : 1));
    for (int64_t n = 0; n < N; ++n) {
      for (int j = 0; j < spd; ++j) prod[j] = loss[1 + j] * noise[(size_t)n * spd + j];
      grad[n] = (fbo_np_sum(prod, spd) / (double)spd) / p->sigma;     /* :244 */
    }
    free(prod);
  }
  free(z); free(aud); free(noise); free(scores); free(loss);
  return rc;
}

static double sgn(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : (x == 0.0 ? 0.0 : x)); }
static double clipd(double x, double lo, double hi) { /* np.clip = minimum(maximum(x,lo),hi) */
  double y = x < lo ? lo : x;
  return y > hi ? hi : y;
}

int fbo_attack(const fbo_nes_params *p, fbo_score_fn fn, void *ctx, const double *audio,
               int64_t N, const double *noise_all, uint64_t seed, uint32_t stream,
               int16_t *adv_i16, double *adver_out, double *trace, int *n_trace) {
  int half = p->samples_per_draw / 2, S = p->n_spk;
  double *adver = (double *)malloc(sizeof(double) * N);
  double *grad = (double *)calloc(N, sizeof(double));   /* grad = 0 :157 */
  double *g = (double *)malloc(sizeof(double) * N);
  double *lower = (double *)malloc(sizeof(double) * N), *upper = (double *)malloc(sizeof(double) * N);
  double *last_ls = (double *)malloc(sizeof(double) * (p->plateau_length + 1));
  double *score0 = (double *)malloc(sizeof(double) * S);
  int n_ls = 0, it = 0, rows = 0, rc = 0;
  double lr = p->max_lr;
  fbo_nes_params q = *p;
  for (int64_t n = 0; n < N; ++n) {
    adver[n] = audio[n];
    lower[n] = clipd(audio[n] - p->epsilon, -1.0, 1.0);               /* :163-164 */
    upper[n] = clipd(audio[n] + p->epsilon, -1.0, 1.0);
  }
  int broke = 0;
  for (it = 0; it < p->max_iter; ++it) {                               /* :168 */
    double loss, adver_loss;
    rc = fbo_get_grad(&q, fn, ctx, adver, N,
                      noise_all ? noise_all + (size_t)it * N * half : NULL, seed, (uint32_t)it,
                      stream, &loss, g, &adver_loss, score0);
    if (rc) break;
    double dist = 0.0;
    for (int64_t n = 0; n < N; ++n) { double d = fabs(audio[n] - adver[n]); if (d > dist) dist = d; }
    double *row = trace ? trace + (size_t)rows * (3 + S) : NULL;
    if (adver_loss < 0.0) {                                            /* :181 */
      if (row) { row[0] = dist; row[1] = adver_loss; row[2] = lr; memcpy(row + 3, score0, sizeof(double) * S); }
      ++rows; broke = 1;
      break;
    }
    for (int64_t n = 0; n < N; ++n)                                    /* :193 */
      grad[n] = p->momentum * grad[n] + (1.0 - p->momentum) * g[n];
    last_ls[n_ls++] = loss;                                            /* :195-200 */
    if (n_ls > p->plateau_length) { memmove(last_ls, last_ls + 1, sizeof(double) * p->plateau_length); n_ls = p->plateau_length; }
    if (last_ls[n_ls - 1] > last_ls[0] && n_ls == p->plateau_length) {
      if (lr > p->min_lr) { double l2 = lr / p->plateau_drop; lr = l2 > p->min_lr ? l2 : p->min_lr; }
      n_ls = 0;
    }
    for (int64_t n = 0; n < N; ++n) {                                  /* :202-203 */
      adver[n] -= lr * sgn(grad[n]);
      adver[n] = clipd(adver[n], lower[n], upper[n]);
    }
    if (row) { row[0] = dist; row[1] = adver_loss; row[2] = lr; memcpy(row + 3, score0, sizeof(double) * S); }
    ++rows;
  }
  int last_iter = broke ? it : p->max_iter - 1; /* python `iter` after the loop */
  int flag = (last_iter < p->max_iter - 1) ? 1 : -1;                   /* :219 */
  if (p->max_iter <= 0) flag = 0;
  fbo_quantize(adver, N, 16, adv_i16);                                 /* :220 */
  if (adver_out) memcpy(adver_out, adver, sizeof(double) * N);
  if (n_trace) *n_trace = rows;
  free(adver); free(grad); free(g); free(lower); free(upper); free(last_ls); free(score0);
  return rc ? 0 : flag;
}

int fbo_estimate_threshold(const fbo_nes_params *p, double model_threshold, fbo_score_fn fn,
                           void *ctx, const double *audio, int64_t N, const double *noise_all,
                           int max_total_iters, uint64_t seed, uint32_t stream,
                           double *score_out, int *n_iters_out, int *n_outer_out,
                           double *thr_final, double *adver_out) {
  if (p->task == FBO_TASK_CSI) return 1;                               /* :41-43 */
  int half = p->samples_per_draw / 2, S = p->n_spk;
  double *sc = (double *)malloc(sizeof(double) * S);
  double *adver = (double *)malloc(sizeof(double) * N);
  double *grad = (double *)calloc(N, sizeof(double));
  double *g = (double *)malloc(sizeof(double) * N);
  double *lower = (double *)malloc(sizeof(double) * N), *upper = (double *)malloc(sizeof(double) * N);
  double *last_ls = (double *)malloc(sizeof(double) * (p->plateau_length + 1));
  int rc = fn(ctx, audio, N, 1, sc);                                   /* :53 */
  double init = sc[0];
  for (int j = 1; j < S; ++j) if (sc[j] > init) init = sc[j];          /* :54-55 */
  double delta = fabs(init / 10.0);                                    /* :57 */
  fbo_nes_params q = *p;
  q.threshold = init + delta;                                          /* :59 */
  q.attack_type = FBO_UNTARGETED;                                      /* :73-74 */
  for (int64_t n = 0; n < N; ++n) {
    adver[n] = audio[n];
    lower[n] = clipd(audio[n] - p->epsilon, -1.0, 1.0);
    upper[n] = clipd(audio[n] + p->epsilon, -1.0, 1.0);
  }
  int n_iters = 0, n_outer = 0, done = 0;
  while (!rc && !done) {                                               /* :76 */
    double lr = p->max_lr;
    int n_ls = 0;
    for (;;) {                                                         /* :85 */
      rc = fn(ctx, adver, N, 1, sc);                                   /* make_decisions :89 */
      if (rc) break;
      double s = sc[0];
      for (int j = 1; j < S; ++j) if (sc[j] > s) s = sc[j];
      if (s >= model_threshold) { *score_out = s; done = 1; break; }   /* decision != -1 :96-103 */
      if (s >= q.threshold) break;                                     /* :105-109 */
      if (n_iters >= max_total_iters) { rc = -2; break; }
      double loss, al;
      rc = fbo_get_grad(&q, fn, ctx, adver, N,
                        noise_all ? noise_all + (size_t)n_iters * N * half : NULL, seed,
                        (uint32_t)n_iters, stream, &loss, g, &al, sc);
      if (rc) break;
      for (int64_t n = 0; n < N; ++n)
        grad[n] = p->momentum * grad[n] + (1.0 - p->momentum) * g[n];  /* :114 */
      last_ls[n_ls++] = loss;
      if (n_ls > p->plateau_length) { memmove(last_ls, last_ls + 1, sizeof(double) * p->plateau_length); n_ls = p->plateau_length; }
      if (last_ls[n_ls - 1] > last_ls[0] && n_ls == p->plateau_length) {
        if (lr > p->min_lr) { double l2 = lr / p->plateau_drop; lr = l2 > p->min_lr ? l2 : p->min_lr; }
        n_ls = 0;
      }
      for (int64_t n = 0; n < N; ++n) {
        adver[n] -= lr * sgn(grad[n]);
        adver[n] = clipd(adver[n], lower[n], upper[n]);
      }
      ++n_iters;
    }
    if (!done && !rc) { q.threshold += delta; ++n_outer; }             /* :135-137 */
  }
  if (n_iters_out) *n_iters_out = n_iters;
  if (n_outer_out) *n_outer_out = n_outer;
  if (thr_final) *thr_final = q.threshold;
  if (adver_out) memcpy(adver_out, adver, sizeof(double) * N);
  free(sc); free(adver); free(grad); free(g); free(lower); free(upper); free(last_ls);
  return rc;
}

/* ------------------------------------------- built-in GMM system scorer */
int fbo_gmm_system_score(void *ctx, const double *audios, int64_t N, int B, double *scores) {
  fbo_gmm_system *g = (fbo_gmm_system *)ctx;
  int16_t *wav = (int16_t *)malloc(sizeof(int16_t) * (size_t)N * B);
  int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (B + 1));
  fbo_quantize(audios, N * B, 16, wav);
  for (int b = 0; b <= B; ++b) off[b] = (int64_t)b * N;
  double *raw = (double *)malloc(sizeof(double) * (size_t)B * g->M);
  int rc = fbo_gmm_score_batch(&g->cfg, wav, off, B, g->gconsts, g->means_invvars, g->inv_vars,
                               g->M, g->C, g->D, raw, NULL, g->nthreads);
  g->scored_utts += B;
  if (rc == 0) {
    if (g->task == FBO_TASK_CSI) {
      for (int b = 0; b < B; ++b)
        for (int m = 0; m < g->M; ++m)
          scores[(size_t)b * g->M + m] = (raw[(size_t)b * g->M + m] - g->z_mean[m]) / g->z_std[m];
    } else {
      int S = g->M - 1;
      for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s)
          scores[(size_t)b * S + s] = raw[(size_t)b * g->M + 1 + s] - raw[(size_t)b * g->M];
    }
  }
  free(wav); free(off); free(raw);
  return rc;
}

/* ======================================================== i-vector / PLDA */
/* [EXT] sid/extract_ivectors.sh: gmm-gselect --n=20 | fgmm-global-gselect-to-post
 * --min-post=0.025 | scale-post 1.0 | ivector-extract  (SURVEY.md A.9). */
typedef struct { float v; int k; } fbo_pair;
static int pair_desc(const void *a, const void *b) {
  const fbo_pair *x = (const fbo_pair *)a, *y = (const fbo_pair *)b;
  if (x->v > y->v) return -1;
  if (x->v < y->v) return 1;
  return (x->k > y->k) ? -1 : (x->k < y->k ? 1 : 0); /* std::greater<pair<float,int>> */
}

void fbo_iv_stats(const fbo_iv_system *s, const float *feats, int Tv, double *gamma, double *X) {
  const int C = s->C, D = s->D, n = s->num_gselect < C ? s->num_gselect : C;
  const int tri = D * (D + 1) / 2;
  fbo_pair *pr = (fbo_pair *)malloc(sizeof(fbo_pair) * C);
  double *xd = (double *)malloc(sizeof(double) * D), *xq = (double *)malloc(sizeof(double) * D);
  double *ll = (double *)malloc(sizeof(double) * n);
  float *post = (float *)malloc(sizeof(float) * n);
  memset(gamma, 0, sizeof(double) * C);
  memset(X, 0, sizeof(double) * (size_t)C * D);
  const float min_post = (float)s->min_post;
  for (int t = 0; t < Tv; ++t) {
    const float *f = feats + (size_t)t * D;
    for (int d = 0; d < D; ++d) { xd[d] = (double)f[d]; xq[d] = (double)(float)(f[d] * f[d]); }
    /* gmm-gselect on the diagonalised UBM (float32 log-likelihoods like Kaldi) */
    for (int k = 0; k < C; ++k) {
      const float *m = s->dg_means_invvars + (size_t)k * D, *v = s->dg_inv_vars + (size_t)k * D;
      double a = 0.0, b = 0.0;
      for (int d = 0; d < D; ++d) { a += (double)m[d] * xd[d]; b += (double)v[d] * xq[d]; }
      pr[k].v = (float)((double)s->dg_gconsts[k] + a - 0.5 * b);
      pr[k].k = k;
    }
    qsort(pr, C, sizeof(fbo_pair), pair_desc);
    /* fgmm-global-gselect-to-post: full-covariance log-likelihoods of the selected Gaussians */
    double mx = -INFINITY;
    for (int j = 0; j < n; ++j) {
      const int k = pr[j].k;
      const float *mic = s->fg_means_invcovars + (size_t)k * D;
      const float *P = s->fg_inv_covars + (size_t)k * tri;
      double lin = 0.0, quad = 0.0;
      for (int r = 0; r < D; ++r) {
        lin += (double)mic[r] * xd[r];
        const float *row = P + (size_t)r * (r + 1) / 2;
        double acc = 0.0;
        for (int c = 0; c < r; ++c) acc += (double)row[c] * xd[c];
        quad += xd[r] * (2.0 * acc + (double)row[r] * xd[r]);
      }
      ll[j] = (double)(float)((double)s->fg_gconsts[k] + lin - 0.5 * quad);
      if (ll[j] > mx) mx = ll[j];
    }
    double sum = 0.0;
    for (int j = 0; j < n; ++j) { ll[j] = exp(ll[j] - mx); sum += ll[j]; }
    int jmax = 0;
    for (int j = 0; j < n; ++j) { post[j] = (float)(ll[j] / sum); if (post[j] > post[jmax]) jmax = j; }
    if (min_post != 0.0f) {
      double s2 = 0.0;
      for (int j = 0; j < n; ++j) { if (post[j] < min_post) post[j] = 0.0f; s2 += (double)post[j]; }
      if (s2 == 0.0) post[jmax] = 1.0f;
      else for (int j = 0; j < n; ++j) post[j] = (float)((double)post[j] / s2);
    }
    for (int j = 0; j < n; ++j) {
      if (post[j] == 0.0f) continue;
      const int k = pr[j].k;
      const double w = (double)post[j];
      gamma[k] += w;
      double *xk = X + (size_t)k * D;
      for (int d = 0; d < D; ++d) xk[d] += w * xd[d];
    }
  }
  free(pr); free(xd); free(xq); free(ll); free(post);
}

/* Cholesky solve of a symmetric positive definite system, A overwritten */
static int chol_solve(double *A, double *b, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return -1;
    d = sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * b[k];
    b[i] = v / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * b[k];
    b[i] = v / A[(size_t)i * n + i];
  }
  return 0;
}

/* IvectorExtractor::GetIvectorDistribution without weight projections:
 * linear = sum_k (Sigma_k^-1 M_k)^T X_k, quadratic = sum_k gamma_k U_k; prior: linear[0] +=
 * prior_offset, quadratic += I; ivec = quadratic^-1 linear; ivec[0] -= prior_offset. */
int fbo_iv_extract(const fbo_iv_system *s, const double *gamma, const double *X, double *ivec) {
  const int C = s->C, D = s->D, R = s->R, tri = R * (R + 1) / 2;
  double *lin = (double *)calloc(R, sizeof(double));
  double *qp = (double *)calloc(tri, sizeof(double));
  for (int k = 0; k < C; ++k) {
    if (gamma[k] == 0.0) continue;
    const double *sm = s->sigma_inv_m + (size_t)k * D * R;
    const double *xk = X + (size_t)k * D;
    for (int d = 0; d < D; ++d) {
      const double xv = xk[d];
      const double *row = sm + (size_t)d * R;
      for (int r = 0; r < R; ++r) lin[r] += row[r] * xv;
    }
    const double g = gamma[k];
    const double *uk = s->u + (size_t)k * tri;
    for (int i = 0; i < tri; ++i) qp[i] += g * uk[i];
  }
  lin[0] += s->prior_offset;
  double *A = (double *)malloc(sizeof(double) * (size_t)R * R);
  for (int r = 0; r < R; ++r)
    for (int c = 0; c <= r; ++c) {
      double v = qp[(size_t)r * (r + 1) / 2 + c] + (r == c ? 1.0 : 0.0);
      A[(size_t)r * R + c] = v;
      A[(size_t)c * R + r] = v;
    }
  int rc = chol_solve(A, lin, R);
  for (int r = 0; r < R; ++r) ivec[r] = lin[r];
  ivec[0] -= s->prior_offset;
  free(lin); free(qp); free(A);
  return rc;
}

/* ivector-subtract-global-mean | transform-vec | ivector-normalize-length, then
 * Plda::TransformIvector with normalize_length=true, simple_length_norm=false, n=1 (A.10) */
void fbo_iv_backend(const fbo_iv_system *s, const double *ivec_in, double *y) {
  const int R = s->R, L = s->L;
  /* ivector-extract writes a float vector */
  double *x = (double *)malloc(sizeof(double) * R), *z = (double *)malloc(sizeof(double) * L);
  for (int r = 0; r < R; ++r) x[r] = (double)(float)ivec_in[r] - s->mean_vec[r];
  double nrm = 0.0;
  for (int l = 0; l < L; ++l) {
    const double *row = s->lda + (size_t)l * s->lda_cols;
    double acc = s->lda_cols == R + 1 ? row[R] : 0.0;
    for (int r = 0; r < R; ++r) acc += row[r] * x[r];
    z[l] = acc;
    nrm += acc * acc;
  }
  nrm = sqrt(nrm);
  const double ratio = nrm / sqrt((double)L);
  if (ratio != 0.0) for (int l = 0; l < L; ++l) z[l] /= ratio;
  double dot = 0.0;
  for (int l = 0; l < L; ++l) {
    const double *row = s->plda_transform + (size_t)l * L;
    double acc = 0.0;
    for (int m = 0; m < L; ++m) acc += row[m] * (z[m] - s->plda_mean[m]);
    y[l] = acc;
    dot += acc * acc / (s->plda_psi[l] + 1.0);
  }
  const double nf = sqrt((double)L / dot);
  for (int l = 0; l < L; ++l) y[l] *= nf;
  free(x); free(z);
}

double fbo_plda_llr(const fbo_iv_system *s, const double *train, const double *y) {
  const int L = s->L;
  double given = 0.0, without = 0.0;
  for (int l = 0; l < L; ++l) {
    const double psi = s->plda_psi[l];
    const double mean = psi / (psi + 1.0) * train[l];
    const double var = 1.0 + psi / (psi + 1.0);
    const double d = y[l] - mean;
    given += log(var) + d * d / var;
    without += log(psi + 1.0) + y[l] * y[l] / (psi + 1.0);
  }
  const double M_LOG_2PI_ = 1.8378770664093454835606594728112;
  given = -0.5 * (given + M_LOG_2PI_ * L);
  without = -0.5 * (without + M_LOG_2PI_ * L);
  return given - without;
}

int fbo_iv_score_batch(const fbo_iv_system *s, const int16_t *wav, const int64_t *off, int B,
                       double *llr, double *ivecs_out, int *tv_out) {
  int err = 0;
  const int D = s->D, C = s->C, R = s->R, L = s->L, S = s->S;
  if (fbo_feat_dim(&s->cfg) != D) return -1000000;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(s->nthreads > 0 ? s->nthreads : 1)
#endif
  for (int b = 0; b < B; ++b) {
    const int64_t n = off[b + 1] - off[b];
    const int T = fbo_num_frames(&s->cfg, n);
    float *feats = (float *)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1) * D);
    int Tt = 0;
    const int tv = fbo_frontend(&s->cfg, wav + off[b], n, feats, &Tt);
    if (tv_out) tv_out[b] = tv;
    if (tv <= 0) {
#ifdef _OPENMP
#pragma omp critical
#endif
      { if (err == 0 || -(b + 1) > err) err = -(b + 1); }
      for (int j = 0; j < S; ++j) llr[(size_t)b * S + j] = NAN;
    } else {
      double *gamma = (double *)malloc(sizeof(double) * C), *X = (double *)malloc(sizeof(double) * (size_t)C * D);
      double *iv = (double *)malloc(sizeof(double) * R), *y = (double *)malloc(sizeof(double) * L);
      fbo_iv_stats(s, feats, tv, gamma, X);
      if (fbo_iv_extract(s, gamma, X, iv) != 0) {
#ifdef _OPENMP
#pragma omp critical
#endif
        { err = -2000000; }
      }
      if (ivecs_out) memcpy(ivecs_out + (size_t)b * R, iv, sizeof(double) * R);
      fbo_iv_backend(s, iv, y);
      for (int j = 0; j < S; ++j) {
        const double v = fbo_plda_llr(s, s->train + (size_t)j * L, y);
        llr[(size_t)b * S + j] = s->cfg.text_scores ? fbo_round6(v) : v;  /* ivector-plda-scoring writes text */
      }
      free(gamma); free(X); free(iv); free(y);
    }
    free(feats);
  }
  return err;
}

int fbo_iv_system_score(void *ctx, const double *audios, int64_t N, int B, double *scores) {
  const fbo_iv_system *s = (const fbo_iv_system *)ctx;
  int16_t *wav = (int16_t *)malloc(sizeof(int16_t) * (size_t)N * B);
  int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (B + 1));
  fbo_quantize(audios, N * B, 16, wav);
  for (int b = 0; b <= B; ++b) off[b] = (int64_t)b * N;
  int rc = fbo_iv_score_batch(s, wav, off, B, scores, NULL, NULL);
  if (rc == 0)
    for (int b = 0; b < B; ++b)
      for (int j = 0; j < s->S; ++j)
        scores[(size_t)b * s->S + j] = (scores[(size_t)b * s->S + j] - s->z_mean[j]) / s->z_std[j];
  free(wav); free(off);
  return rc;
}
