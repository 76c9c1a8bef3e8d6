// ivector_kernels.hip -- K8..K12: the i-vector / PLDA scoring back-end on gfx950.
//
// Replaces `sid/extract_ivectors.sh` (gmm-gselect --n=20 | fgmm-global-gselect-to-post
// --min-post=0.025 | scale-post | ivector-extract) and `ivector-plda-scoring` with the
// `ivector-subtract-global-mean | transform-vec | ivector-normalize-length` pipes the reference
// launches per scoring call (ivector_PLDA_kaldiHelper.py:197-213, 251-280); algorithms from
// SURVEY.md A.9/A.10 ([EXT]).  Extractor / PLDA arithmetic is float64 like Kaldi's; posteriors
// are float32 values like Kaldi's, accumulated into float64 statistics in frame order
// (deterministic: no atomics anywhere).
//
//   k_gmm<KH,DUMP>      diagonalised-UBM log-likelihoods of every component (gmm_kernels.hip)
//   k_iv_select_post    per frame: top-n Gaussians, full-covariance log-likelihoods, softmax,
//                       min-post pruning
//   k_iv_stats          zeroth / first order statistics per (utterance, 64-component slab)
//   k_iv_lin / _quad    the T-matrix contraction: lin = sum_k (S_k^-1 M_k)^T F_k,
//                       quad = I + sum_k N_k U_k   (HBM-streaming of 0.47 GB + 1.3 GB of float64)
//   k_iv_solve          blocked Cholesky + triangular solves of the B (R x R) systems
//   k_iv_backend        mean subtraction, LDA, length norm, PLDA transform, LLR vs enrolled
#include <float.h>

#include "fb_device.h"
#include "fb_kernels.h"

// ------------------------------------------------------- derived variables
// SIM[k][d][r] = sum_e Sinv[k][d][e] * M[k][e][r]
__global__ __launch_bounds__(256) void k_iv_derive_sim(int C, int D, int R, const double *__restrict__ M,
                                                       const double *__restrict__ sinv_packed,
                                                       double *__restrict__ sim) {
  const int k = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= D * R) return;
  const int d = idx / R, r = idx - d * R;
  const double *P = sinv_packed + (size_t)k * (D * (D + 1) / 2);
  const double *Mk = M + (size_t)k * D * R;
  double acc = 0.0;
  for (int e = 0; e < D; ++e) {
    const double pv = (e <= d) ? P[(size_t)d * (d + 1) / 2 + e] : P[(size_t)e * (e + 1) / 2 + d];
    acc = fma(pv, Mk[(size_t)e * R + r], acc);
  }
  sim[(size_t)k * D * R + idx] = acc;
}
// U[k][tri(i,j)] = sum_d M[k][d][i] * SIM[k][d][j],  j <= i
__global__ __launch_bounds__(256) void k_iv_derive_u(int C, int D, int R, const double *__restrict__ M,
                                                     const double *__restrict__ sim, double *__restrict__ u) {
  const int k = blockIdx.y;
  const int triR = R * (R + 1) / 2;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= triR) return;
  int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= e) ++i;
  while (i * (i + 1) / 2 > e) --i;
  const int j = e - i * (i + 1) / 2;
  const double *Mk = M + (size_t)k * D * R, *Sk = sim + (size_t)k * D * R;
  double acc = 0.0;
  for (int d = 0; d < D; ++d) acc = fma(Mk[(size_t)d * R + i], Sk[(size_t)d * R + j], acc);
  u[(size_t)k * triR + e] = acc;
}
void fb_launch_iv_derive(hipStream_t s, int C, int D, int R, const double *M, const double *sinv_packed,
                         double *sim, double *u) {
  hipLaunchKernelGGL(k_iv_derive_sim, dim3((D * R + 255) / 256, C), dim3(256), 0, s, C, D, R, M, sinv_packed, sim);
  const int triR = R * (R + 1) / 2;
  hipLaunchKernelGGL(k_iv_derive_u, dim3((triR + 255) / 256, C), dim3(256), 0, s, C, D, R, M, sim, u);
}

// ------------------------------------------------ gselect + posteriors (K8/K9)
// One wave per voiced frame.  LDS per wave: the Cpad log-likelihoods of the frame + its D features.
__global__ __launch_bounds__(256) void k_iv_select_post(FbIvDev iv, const float *__restrict__ ll,
                                                        const float *__restrict__ feats,
                                                        const int *__restrict__ n_rows_ptr,
                                                        int *__restrict__ sel, float *__restrict__ post) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n_rows = *n_rows_ptr;
  const int row = blockIdx.x * 4 + w;
  if (row >= n_rows) return;  // whole wave exits; no block-level barriers below
  const int Cpad = iv.Cpad, D = iv.D, nsel = iv.nsel;
  float *vals = smf + (size_t)w * (Cpad + 128);
  float *xs = vals + Cpad;
  const float *lr = ll + (size_t)row * Cpad;
  for (int i = lane; i < Cpad; i += 64) vals[i] = (i < iv.C) ? lr[i] : -FLT_MAX;
  for (int i = lane; i < D; i += 64) xs[i] = feats[(size_t)row * D + i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- top-nsel, descending by (value, index) like std::greater<pair<float,int>>
  int my_k = -1;  // lane j < nsel keeps the j-th selected component
  for (int s = 0; s < nsel; ++s) {
    float bv = -FLT_MAX;
    int bi = -1;
    for (int i = lane; i < Cpad; i += 64) {
      const float v = vals[i];
      if (v > bv || (v == bv && i > bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
    }
    if (lane == s) my_k = bi;
    if (lane == 0 && bi >= 0) vals[bi] = -FLT_MAX;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // ---- full-covariance log-likelihoods of the selected components
  const int triD = iv.triD;
  double my_ll = -INFINITY;
  for (int s = 0; s < nsel; ++s) {
    const int k = __shfl(my_k, s, 64);
    double acc = 0.0;
    if (k >= 0 && k < iv.C) {
      const float *P = iv.fg_P + (size_t)k * triD;
      const float *mic = iv.fg_mic + (size_t)k * D;
      for (int e = lane; e < triD; e += 64) {
        const int r = iv.tri_r[e], c = iv.tri_c[e];
        const double xr = (double)xs[r], xc = (double)xs[c];
        const double t = (double)P[e] * xr * xc;
        acc -= (r == c) ? 0.5 * t : t;
      }
      for (int d = lane; d < D; d += 64) acc = fma((double)mic[d], (double)xs[d], acc);
    }
    acc = fb_wave_sum(acc);
    if (lane == s) my_ll = (k >= 0 && k < iv.C) ? (double)(float)((double)iv.fg_gconsts[k] + acc) : -INFINITY;
  }
  // ---- softmax over the nsel lanes, min-post pruning, renormalisation
  const double mx = fb_wave_max(lane < nsel ? my_ll : -INFINITY);
  double ex = (lane < nsel && my_ll > -INFINITY) ? exp(my_ll - mx) : 0.0;
  const double sum = fb_wave_sum(ex);
  float p = (lane < nsel) ? (float)(ex / sum) : 0.0f;
  const float min_post = iv.min_post;
  if (min_post != 0.0f) {
    // first lane holding the maximum posterior (Vector::Max(&index) semantics: first max)
    float pm = p;
    int pi = lane < nsel ? lane : 64;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(pm, o, 64);
      const int oi = __shfl_xor(pi, o, 64);
      if (ov > pm || (ov == pm && oi < pi)) { pm = ov; pi = oi; }
    }
    if (p < min_post) p = 0.0f;
    const double s2 = fb_wave_sum((double)p);
    if (s2 == 0.0) p = (lane == pi) ? 1.0f : 0.0f;
    else p = (float)((double)p / s2);
  }
  if (lane < nsel) {
    sel[(size_t)row * nsel + lane] = my_k;
    post[(size_t)row * nsel + lane] = p;
  }
}
void fb_launch_iv_select_post(hipStream_t s, const FbIvDev &iv, const float *ll, const float *feats,
                              const int *n_rows_ptr, int rows_cap, int *sel, float *post) {
  if (rows_cap <= 0) return;
  size_t shm = sizeof(float) * 4 * (size_t)(iv.Cpad + 128);
  hipLaunchKernelGGL(k_iv_select_post, dim3((rows_cap + 3) / 4), dim3(256), shm, s, iv, ll, feats, n_rows_ptr, sel,
                     post);
}

// ------------------------------------------------------- statistics (K10a)
// grid (B, C/64): thread (c = tid&63, dg = tid>>6) owns component k0+c and feature dims dg, dg+4, ...
#define FB_IV_DMAX4 20  // D <= 80
__global__ __launch_bounds__(256) void k_iv_stats(FbIvDev iv, const float *__restrict__ feats,
                                                  const int *__restrict__ row_off, const int *__restrict__ sel,
                                                  const float *__restrict__ post, double *__restrict__ gamma,
                                                  double *__restrict__ X) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int b = blockIdx.x, k0 = blockIdx.y * 64;
  const int D = iv.D, nsel = iv.nsel;
  float *Pd = smf;           // [64 rows][64 comps]
  float *F = smf + 64 * 64;  // [64 rows][D]
  const int r0 = row_off[b], r1 = row_off[b + 1];
  const int c = threadIdx.x & 63, dg = threadIdx.x >> 6;
  double acc[FB_IV_DMAX4];
#pragma unroll
  for (int i = 0; i < FB_IV_DMAX4; ++i) acc[i] = 0.0;
  double gam = 0.0;
  for (int rb = r0; rb < r1; rb += 64) {
    const int nr = min(64, r1 - rb);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) Pd[i] = 0.0f;
    __syncthreads();
    for (int e = threadIdx.x; e < nr * nsel; e += 256) {
      const int rl = e / nsel;
      const size_t o = (size_t)(rb + rl) * nsel + (e - rl * nsel);
      const float p = post[o];
      const int k = sel[o] - k0;
      if (p != 0.0f && k >= 0 && k < 64) Pd[rl * 64 + k] = p;
    }
    for (int i = threadIdx.x; i < nr * D; i += 256) F[i] = feats[(size_t)rb * D + i];
    __syncthreads();
    for (int rl = 0; rl < nr; ++rl) {
      const float p = Pd[rl * 64 + c];
      if (p != 0.0f) {
        const double wv = (double)p;
        gam = __dadd_rn(gam, wv);
        const float *fr = F + rl * D;
#pragma unroll
        for (int i = 0; i < FB_IV_DMAX4; ++i) {
          const int d = dg + 4 * i;
          if (d < D) acc[i] = __dadd_rn(acc[i], __dmul_rn(wv, (double)fr[d]));
        }
      }
    }
    __syncthreads();
  }
  const int k = k0 + c;
  if (k < iv.C) {
    if (dg == 0) gamma[(size_t)b * iv.C + k] = gam;
#pragma unroll
    for (int i = 0; i < FB_IV_DMAX4; ++i) {
      const int d = dg + 4 * i;
      if (d < D) X[((size_t)b * iv.C + k) * D + d] = acc[i];
    }
  }
}
void fb_launch_iv_stats(hipStream_t s, const FbIvDev &iv, const float *feats, const int *row_off, const int *sel,
                        const float *post, int B, double *gamma, double *X) {
  size_t shm = sizeof(float) * (64 * 64 + 64 * (size_t)iv.D);
  hipLaunchKernelGGL(k_iv_stats, dim3(B, (iv.C + 63) / 64), dim3(256), shm, s, iv, feats, row_off, sel, post, gamma,
                     X);
}

// ---------------------------------------------- T-matrix contraction (K10b)
// lin partials: grid (n_kchunks, ceil(B/BT)); thread = one i-vector dimension r.
#define FB_IV_BT 32
__global__ __launch_bounds__(512) void k_iv_lin(FbIvDev iv, const double *__restrict__ X, int B, int rows_per_chunk,
                                                double *__restrict__ linp) {
  extern __shared__ __attribute__((aligned(16))) double smd[];  // [BT][32]
  const int R = iv.R;
  const int64_t Q = (int64_t)iv.C * iv.D;
  const int64_t q0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t q1 = min(Q, q0 + rows_per_chunk);
  const int b0 = blockIdx.y * FB_IV_BT;
  const int nb = min(FB_IV_BT, B - b0);
  const int r = threadIdx.x;
  double acc[FB_IV_BT];
#pragma unroll
  for (int i = 0; i < FB_IV_BT; ++i) acc[i] = 0.0;
  for (int64_t qb = q0; qb < q1; qb += 32) {
    const int nq = (int)min((int64_t)32, q1 - qb);
    __syncthreads();
    for (int i = threadIdx.x; i < FB_IV_BT * 32; i += blockDim.x) {
      const int bb = i >> 5, qq = i & 31;
      smd[i] = (bb < nb && qq < nq) ? X[(size_t)(b0 + bb) * Q + qb + qq] : 0.0;
    }
    __syncthreads();
    if (r < R) {
      for (int qq = 0; qq < nq; ++qq) {
        const double v = iv.sim[(size_t)(qb + qq) * R + r];
#pragma unroll
        for (int bb = 0; bb < FB_IV_BT; ++bb) acc[bb] = fma(smd[bb * 32 + qq], v, acc[bb]);
      }
    }
  }
  if (r < R)
    for (int bb = 0; bb < nb; ++bb) linp[((size_t)blockIdx.x * B + b0 + bb) * R + r] = acc[bb];
}
// quad[b][e] = sum_k gamma[b][k] U[k][e]; thread = one packed element e; BT utterances per pass
template <int BT>
__global__ __launch_bounds__(256) void k_iv_quad(FbIvDev iv, const double *__restrict__ gamma, int B,
                                                 double *__restrict__ quad) {
  __shared__ double sg[BT * 64];
  const int triR = iv.triR, C = iv.C;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * BT;
  const int nb = min(BT, B - b0);
  double acc[BT];
#pragma unroll
  for (int i = 0; i < BT; ++i) acc[i] = 0.0;
  for (int kb = 0; kb < C; kb += 64) {
    const int nk = min(64, C - kb);
    __syncthreads();
    for (int i = threadIdx.x; i < BT * 64; i += 256) {
      const int bb = i >> 6, kk = i & 63;
      sg[i] = (bb < nb && kk < nk) ? gamma[(size_t)(b0 + bb) * C + kb + kk] : 0.0;
    }
    __syncthreads();
    if (e < triR) {
      for (int kk = 0; kk < nk; ++kk) {
        const double uv = iv.u[(size_t)(kb + kk) * triR + e];
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) acc[bb] = fma(sg[bb * 64 + kk], uv, acc[bb]);
      }
    }
  }
  if (e < triR)
    for (int bb = 0; bb < nb; ++bb) quad[(size_t)(b0 + bb) * triR + e] = acc[bb];
}
void fb_launch_iv_contract(hipStream_t s, const FbIvDev &iv, const double *gamma, const double *X, int B,
                           int n_kchunks, double *linp, double *quad) {
  const int64_t Q = (int64_t)iv.C * iv.D;
  int rpc = (int)((Q + n_kchunks - 1) / n_kchunks);
  rpc = (rpc + 31) / 32 * 32;
  const int threads = (iv.R + 63) / 64 * 64;
  hipLaunchKernelGGL(k_iv_lin, dim3(n_kchunks, (B + FB_IV_BT - 1) / FB_IV_BT), dim3(threads),
                     sizeof(double) * FB_IV_BT * 32, s, iv, X, B, rpc, linp);
  if (B > 16)
    hipLaunchKernelGGL((k_iv_quad<64>), dim3((iv.triR + 255) / 256, (B + 63) / 64), dim3(256), 0, s, iv, gamma, B, quad);
  else
    hipLaunchKernelGGL((k_iv_quad<16>), dim3((iv.triR + 255) / 256, (B + 15) / 16), dim3(256), 0, s, iv, gamma, B, quad);
}

// --------------------------------------------------------- solve (K10c)
// One workgroup per utterance: A = I + unpack(quad), rhs = sum of lin partials (+ prior offset);
// right-looking blocked Cholesky (panel 32) in global scratch (L2 resident), then the two
// triangular solves; ivec = solution with the prior offset removed from component 0.
#define FB_IV_NB 32
__global__ __launch_bounds__(1024) void k_iv_solve(FbIvDev iv, const double *__restrict__ quad,
                                                   const double *__restrict__ linp, int n_kchunks, int B,
                                                   double *__restrict__ Aall, double *__restrict__ ivec,
                                                   int *__restrict__ fail) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  const int R = iv.R, b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  double *A = Aall + (size_t)b * R * R;
  double *rhs = smd;                 // [R]
  double *Dg = rhs + ((R + 1) & ~1); // [NB][NB+1]
  double *Lp = Dg + FB_IV_NB * (FB_IV_NB + 1);  // [R][NB+1] panel below the diagonal block
  const double *qb = quad + (size_t)b * iv.triR;
  for (int i = tid; i < R * R; i += nt) {
    const int r = i / R, c = i - r * R;
    A[i] = (c <= r) ? qb[(size_t)r * (r + 1) / 2 + c] + (r == c ? 1.0 : 0.0) : 0.0;
  }
  for (int r = tid; r < R; r += nt) {
    double acc = 0.0;
    for (int ch = 0; ch < n_kchunks; ++ch) acc += linp[((size_t)ch * B + b) * R + r];
    rhs[r] = acc + (r == 0 ? iv.prior_offset : 0.0);
  }
  __syncthreads();
  for (int j0 = 0; j0 < R; j0 += FB_IV_NB) {
    const int nb = min(FB_IV_NB, R - j0);
    // (a) diagonal block -> LDS, factor with one wave
    for (int i = tid; i < nb * nb; i += nt) {
      const int r = i / nb, c = i - r * nb;
      Dg[r * (FB_IV_NB + 1) + c] = (c <= r) ? A[(size_t)(j0 + r) * R + j0 + c] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      for (int c = 0; c < nb; ++c) {
        const double d = Dg[c * (FB_IV_NB + 1) + c];
        if (!(d > 0.0) && tid == 0) atomicMax(fail, b + 1);
        const double piv = sqrt(d > 0.0 ? d : 1.0);
        for (int r = c + tid; r < nb; r += 64) {
          const double v = (r == c) ? piv : Dg[r * (FB_IV_NB + 1) + c] / piv;
          Dg[r * (FB_IV_NB + 1) + c] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = tid; i < (nb - c - 1) * (nb - c - 1); i += 64) {
          const int rr = c + 1 + i / (nb - c - 1), cc = c + 1 + i % (nb - c - 1);
          if (cc <= rr) Dg[rr * (FB_IV_NB + 1) + cc] -= Dg[rr * (FB_IV_NB + 1) + c] * Dg[cc * (FB_IV_NB + 1) + c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
    for (int i = tid; i < nb * nb; i += nt) {
      const int r = i / nb, c = i - r * nb;
      if (c <= r) A[(size_t)(j0 + r) * R + j0 + c] = Dg[r * (FB_IV_NB + 1) + c];
    }
    // (b) panel below: L21 = A21 * L11^-T, one row per thread
    const int m = R - j0 - nb;
    for (int i = tid; i < m; i += nt) {
      double x[FB_IV_NB];
      const double *arow = A + (size_t)(j0 + nb + i) * R + j0;
#pragma unroll
      for (int c = 0; c < FB_IV_NB; ++c) x[c] = (c < nb) ? arow[c] : 0.0;
#pragma unroll
      for (int c = 0; c < FB_IV_NB; ++c) {
        if (c < nb) {
          double v = x[c];
#pragma unroll
          for (int q = 0; q < FB_IV_NB; ++q)
            if (q < c) v -= x[q] * Dg[c * (FB_IV_NB + 1) + q];
          x[c] = v / Dg[c * (FB_IV_NB + 1) + c];
        }
      }
      double *lrow = Lp + (size_t)i * (FB_IV_NB + 1);
      double *wrow = A + (size_t)(j0 + nb + i) * R + j0;
#pragma unroll
      for (int c = 0; c < FB_IV_NB; ++c)
        if (c < nb) { lrow[c] = x[c]; wrow[c] = x[c]; }
    }
    __syncthreads();
    // (c) trailing update A22 -= L21 L21^T (lower triangle), 2x2 register tiles
    const int mt = (m + 1) / 2;
    for (int i = tid; i < mt * mt; i += nt) {
      const int tr = i / mt, tc = i - tr * mt;
      if (tc > tr) continue;
      const int r0 = 2 * tr, c0 = 2 * tc;
      const double *l0 = Lp + (size_t)r0 * (FB_IV_NB + 1), *l1 = l0 + (FB_IV_NB + 1);
      const double *k0 = Lp + (size_t)c0 * (FB_IV_NB + 1), *k1 = k0 + (FB_IV_NB + 1);
      const bool r1ok = r0 + 1 < m, c1ok = c0 + 1 < m;
      double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
      for (int q = 0; q < nb; ++q) {
        const double a0 = l0[q], a1 = r1ok ? l1[q] : 0.0, b0 = k0[q], b1 = c1ok ? k1[q] : 0.0;
        s00 = fma(a0, b0, s00); s01 = fma(a0, b1, s01); s10 = fma(a1, b0, s10); s11 = fma(a1, b1, s11);
      }
      double *a = A + (size_t)(j0 + nb + r0) * R + j0 + nb + c0;
      a[0] -= s00;
      if (c1ok && c0 + 1 <= r0) a[1] -= s01;
      if (r1ok) {
        a[R] -= s10;
        if (c1ok) a[R + 1] -= s11;
      }
    }
    __syncthreads();
  }
  // ---- forward substitution L y = rhs, then L^T x = y (column oriented, panel by panel)
  for (int j0 = 0; j0 < R; j0 += FB_IV_NB) {
    const int nb = min(FB_IV_NB, R - j0);
    if (tid == 0) {
      for (int c = 0; c < nb; ++c) {
        double v = rhs[j0 + c];
        for (int q = 0; q < c; ++q) v -= A[(size_t)(j0 + c) * R + j0 + q] * rhs[j0 + q];
        rhs[j0 + c] = v / A[(size_t)(j0 + c) * R + j0 + c];
      }
    }
    __syncthreads();
    for (int i = j0 + nb + tid; i < R; i += nt) {
      double v = rhs[i];
      const double *arow = A + (size_t)i * R + j0;
      for (int q = 0; q < nb; ++q) v -= arow[q] * rhs[j0 + q];
      rhs[i] = v;
    }
    __syncthreads();
  }
  for (int j1 = R; j1 > 0; j1 -= FB_IV_NB) {
    const int j0 = max(0, j1 - FB_IV_NB), nb = j1 - j0;
    if (tid == 0) {
      for (int c = nb - 1; c >= 0; --c) {
        double v = rhs[j0 + c];
        for (int q = c + 1; q < nb; ++q) v -= A[(size_t)(j0 + q) * R + j0 + c] * rhs[j0 + q];
        rhs[j0 + c] = v / A[(size_t)(j0 + c) * R + j0 + c];
      }
    }
    __syncthreads();
    for (int i = tid; i < j0; i += nt) {
      double v = rhs[i];
      for (int q = 0; q < nb; ++q) v -= A[(size_t)(j0 + q) * R + i] * rhs[j0 + q];
      rhs[i] = v;
    }
    __syncthreads();
  }
  for (int r = tid; r < R; r += nt) ivec[(size_t)b * R + r] = rhs[r] - (r == 0 ? iv.prior_offset : 0.0);
}
void fb_launch_iv_solve(hipStream_t s, const FbIvDev &iv, const double *quad, const double *linp, int n_kchunks,
                        int B, double *Aall, double *ivec, int *fail) {
  const int R = iv.R;
  size_t shm = sizeof(double) * (((R + 1) & ~1) + FB_IV_NB * (FB_IV_NB + 1) + (size_t)R * (FB_IV_NB + 1));
  hipLaunchKernelGGL(k_iv_solve, dim3(B), dim3(1024), shm, s, iv, quad, linp, n_kchunks, B, Aall, ivec, fail);
}

// ------------------------------------------------------ back-end (K11/K12)
__device__ __forceinline__ double fb_block_sum(double v, double *red) {
  v = fb_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += red[w];
  return r;
}
__global__ __launch_bounds__(256) void k_iv_backend(FbIvDev iv, const double *__restrict__ ivec,
                                                    double *__restrict__ llr) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  __shared__ double red[4];
  const int R = iv.R, L = iv.L, S = iv.S, b = blockIdx.x, tid = threadIdx.x;
  double *x = smd, *z = x + R, *y = z + L;
  for (int r = tid; r < R; r += 256) x[r] = (double)(float)ivec[(size_t)b * R + r] - iv.mean_vec[r];
  __syncthreads();
  double nrm = 0.0;
  for (int l = tid; l < L; l += 256) {
    double acc = iv.lda_cols == R + 1 ? iv.ldaT[(size_t)R * L + l] : 0.0;
    for (int r = 0; r < R; ++r) acc = fma(iv.ldaT[(size_t)r * L + l], x[r], acc);
    z[l] = acc;
    nrm = fma(acc, acc, nrm);
  }
  nrm = sqrt(fb_block_sum(nrm, red));
  const double ratio = nrm / sqrt((double)L);
  for (int l = tid; l < L; l += 256) z[l] = (ratio != 0.0 ? z[l] / ratio : z[l]) - iv.plda_mean[l];
  __syncthreads();
  double dot = 0.0;
  for (int l = tid; l < L; l += 256) {
    double acc = 0.0;
    for (int m = 0; m < L; ++m) acc = fma(iv.pldaT[(size_t)m * L + l], z[m], acc);
    y[l] = acc;
    dot += acc * acc / (iv.plda_psi[l] + 1.0);
  }
  dot = fb_block_sum(dot, red);
  const double nf = sqrt((double)L / dot);
  __syncthreads();
  for (int l = tid; l < L; l += 256) y[l] *= nf;
  __syncthreads();
  const double LOG2PI = 1.8378770664093454835606594728112;
  for (int s = 0; s < S; ++s) {
    const double *tr = iv.train + (size_t)s * L;
    double given = 0.0, without = 0.0;
    for (int l = tid; l < L; l += 256) {
      const double psi = iv.plda_psi[l];
      const double mean = psi / (psi + 1.0) * tr[l];
      const double var = 1.0 + psi / (psi + 1.0);
      const double d = y[l] - mean;
      given += log(var) + d * d / var;
      without += log(psi + 1.0) + y[l] * y[l] / (psi + 1.0);
    }
    given = fb_block_sum(given, red);
    without = fb_block_sum(without, red);
    if (tid == 0)
      llr[(size_t)b * S + s] = -0.5 * (given + LOG2PI * L) - (-0.5 * (without + LOG2PI * L));
  }
}
void fb_launch_iv_backend(hipStream_t s, const FbIvDev &iv, const double *ivec, int B, double *llr) {
  size_t shm = sizeof(double) * (size_t)(iv.R + 2 * iv.L);
  hipLaunchKernelGGL(k_iv_backend, dim3(B), dim3(256), shm, s, iv, ivec, llr);
}
