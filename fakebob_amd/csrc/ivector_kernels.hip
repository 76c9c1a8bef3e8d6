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
//   k_gmm_bx3<NK,true>    diagonalised-UBM log-likelihoods of every component (gmm_kernels.hip)
//   k_iv_select           per frame: top-n Gaussians (registers + DPP arg-max)
//   k_iv_bucket_*         stable, atomic-free partition of the (frame, slot) pairs by component
//   k_iv_fullcov_lds      full-covariance log-likelihoods, component record broadcast from LDS
//   k_iv_post             softmax + min-post pruning per frame
//   k_iv_stats            zeroth / first order statistics per (component, utterance) from the buckets
//   k_iv_active           list of the components with posterior mass
//   k_iv_contract_gemm    the T-matrix contraction: lin = sum_k (S_k^-1 M_k)^T F_k, quad = sum_k N_k U_k
//                         as an LDS-tiled float64-MFMA GEMM over the active rows (of 0.47 GB + 1.3 GB)
//   k_iv_solve_ll         (ivector_solve.hip) blocked left-looking Cholesky + triangular solves of the B (R x R) systems
//   k_iv_backend          mean subtraction, LDA, length norm, PLDA transform, LLR vs enrolled
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fb_device.h"
#include "fb_kernels.h"
#include "fb_iv_tail.h"

typedef double fb_d4 __attribute__((ext_vector_type(4)));  // accumulator of v_mfma_f64_16x16x4_f64

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
// Gathering 20 packed 72x72 precision matrices (10.5 KB each) per frame from L2/MALL is what a
// frame-major kernel spends its time on (3.3 GB per NES batch, measured 1.5 ms).  The work is
// therefore transposed: frames are bucketed by selected component, one workgroup owns one
// component (its matrix lives in registers) and walks the frames that selected it.
//   k_iv_select        wave = frame: top-n of the diagonal log-likelihoods (+ bucket histogram)
//   k_iv_bucket_scan   bucket offsets and the (component, 128-entry chunk) work list
//   k_iv_bucket_fill   (frame, slot) pairs into their bucket
//   k_iv_fullcov       workgroup = (component, chunk): full-covariance log-likelihood of each pair
//   k_iv_post          wave = frame: softmax over the n values, min-post pruning, renormalisation
// Values are independent of the (atomic) bucket order: each pair writes its own output slot.
#define FB_IV_CH 128  // bucket entries per workgroup

// wave = frame, every lane keeps Cpad/64 of the frame's values in REGISTERS (component lane + 64 j) with
// their running maximum; a round = wave arg-max of the 64 lane maxima by DPP (no LDS, no bpermute),
// then only the owning lane removes the winner and rescans its registers.  Order: descending by
// (value, index) like std::greater<pair<float,int>> (gmm-gselect).
// (Round 5 built the alternative the round-4 review asked for -- no dump: every lane of the matrix-core kernel keeps a
//  sorted list of its 16 largest values as 32-bit keys (value with its index in the low significand bits: an insertion
//  is a chain of 16 v_med3_f32), 128 - 256 keys per frame instead of 2 048 values, and a merge kernel recomputes the ~22
//  candidates around the 20th key exactly (float64 sums in the oracle's order) and selects on those; exact by
//  construction, lists of 4 and of 16 gave bit-identical i-vectors, parity green -- and measured it at configs[2] size:
//  k_gmm_fx2 with the lists 64.7 us (four chunks) / 73.0 (eight) against the dump's 51.4 -- some lane of the 64 inserts
//  in nearly every round, 16 + 16 ln(n / 16) insertions per lane are ~100 rounds per chunk --, the merge 80 - 123 us
//  against this kernel's 45 (20 wave-maximum rounds over the keys, ~22 scattered 576-byte parameter rows per frame,
//  20 arg-max rounds): 153 us against 97.  Removed; the dump stays.  Also built and removed: this kernel with TWO frames per
//  wave (a frame over a half-wave, 64 values per lane, every round's instructions serving two frames): bit-identical, and
//  84.9 us against 45 -- at 141 registers three waves per SIMD no longer hide the 16 KB a wave waits for.)
template <int NJ>
__global__ __launch_bounds__(256) void k_iv_select(FbIvDev iv, const float *__restrict__ ll,
                                                   const int *__restrict__ n_rows_ptr, int *__restrict__ sel,
                                                   const int *__restrict__ gate) {
  if (gate && *gate == 0) return;   // the rescue launch behind fb_launch_gsel: no list overflowed, sel[] is complete
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n_rows = *n_rows_ptr;
  const int row = blockIdx.x * 4 + w;
  if (row >= n_rows) return;
  const int nsel = iv.nsel;
  const float *lr = ll + (size_t)row * iv.Cpad;
  float v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = lane + 64 * j;
    v[j] = (i < iv.C) ? lr[min(i, iv.Cpad - 1)] : -FLT_MAX;
  }
  // per-lane candidates: the three best of the lane's registers, refilled (rarely) by a rescan
  float t0v, t1v, t2v;
  int t0j, t1j, t2j;
  // values already handed out are not erased: a rescan only looks at entries that come after the last popped one in
  // the (value descending, index descending) order -- (x, j) < (lim_v, lim_j)
  float lim_v = FLT_MAX;
  int lim_j = NJ;
#define FB_TOP3()                                                                                  \
  {                                                                                                \
    t0v = t1v = t2v = -FLT_MAX;                                                                    \
    t0j = t1j = t2j = -1;                                                                          \
    _Pragma("unroll") for (int j = NJ - 1; j >= 0; --j) { /* descending index + strict '>': ties keep the larger index */ \
      const float x0 = v[j];                                                                       \
      const float x = (x0 < lim_v || (x0 == lim_v && j < lim_j)) ? x0 : -FLT_MAX;                  \
      const bool g0 = x > t0v, g1 = x > t1v, g2 = x > t2v;                                         \
      t2v = g1 ? t1v : (g2 ? x : t2v); t2j = g1 ? t1j : (g2 ? j : t2j);                            \
      t1v = g0 ? t0v : (g1 ? x : t1v); t1j = g0 ? t0j : (g1 ? j : t1j);                            \
      t0v = g0 ? x : t0v; t0j = g0 ? j : t0j;                                                      \
    }                                                                                              \
  }
  FB_TOP3()
  int my_k = -1;  // lane s < nsel keeps the s-th selected component
  for (int s = 0; s < nsel; ++s) {
    float bv = t0v;
    int bi = lane + 64 * (t0j < 0 ? 0 : t0j);
    // fast path: wave maximum of the VALUE alone (6 DPP steps of one v_max each); if exactly one lane holds it, that
    // lane's candidate wins.  Equal maxima in two lanes (or an exhausted wave) take the full (value, index) arg-max.
    float mv = bv;
#define FB_MAX_STEP(CTRL, RM) \
    mv = fmaxf(mv, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(mv), __float_as_int(mv), CTRL, RM, 0xf, false)));
    FB_MAX_STEP(0xb1, 0xf)
    FB_MAX_STEP(0x4e, 0xf)
    FB_MAX_STEP(0x141, 0xf)
    FB_MAX_STEP(0x140, 0xf)
    FB_MAX_STEP(0x142, 0xa)
    FB_MAX_STEP(0x143, 0xc)
#undef FB_MAX_STEP
    const float top = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mv), 63));
    const unsigned long long holders = __ballot(bv == top);
    bool none;
    int win;
    if (__popcll(holders) == 1) {
      none = false;
      win = __builtin_amdgcn_readlane(bi, __ffsll((long long)holders) - 1);
    } else {
#define FB_ARGMAX_STEP(CTRL, RM)                                                              \
    {                                                                                         \
      const float ov = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bv), __float_as_int(bv), CTRL, RM, 0xf, false)); \
      const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, RM, 0xf, false);               \
      if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }                             \
    }
      FB_ARGMAX_STEP(0xb1, 0xf)   // quad_perm [1,0,3,2]
      FB_ARGMAX_STEP(0x4e, 0xf)   // quad_perm [2,3,0,1]
      FB_ARGMAX_STEP(0x141, 0xf)  // row_half_mirror
      FB_ARGMAX_STEP(0x140, 0xf)  // row_mirror
      FB_ARGMAX_STEP(0x142, 0xa)  // row_bcast15
      FB_ARGMAX_STEP(0x143, 0xc)  // row_bcast31
#undef FB_ARGMAX_STEP
      // (fewer than nsel components: the remaining slots stay -1, as with an exhausted heap)
      none = __builtin_amdgcn_readlane(__float_as_int(bv), 63) == __float_as_int(-FLT_MAX);
      win = none ? -1 : __builtin_amdgcn_readlane(bi, 63);
    }
    if (lane == s) my_k = win;
    bool refill = false;
    if (!none && (win & 63) == lane) {  // the owner hands its best candidate out (and forgets the value)
      lim_v = t0v;
      lim_j = t0j;
      t0v = t1v; t0j = t1j; t1v = t2v; t1j = t2j; t2v = -FLT_MAX; t2j = -1;
      refill = t0j < 0;  // candidates exhausted: look at the registers again
    }
    if (__any(refill)) {
      if (refill) FB_TOP3()
    }
  }
#undef FB_TOP3
  if (lane < nsel) sel[(size_t)row * nsel + lane] = my_k;
}

// Stable partition of the (frame, slot) pairs by component -- a counting sort without global atomics:
//   k_iv_bucket_count  block = FB_IV_FB consecutive frames: per-component counts of the block (LDS)
//   k_iv_bucket_scan   per component: exclusive scan over the blocks (-> the block's first slot in the
//                      bucket); totals -> bstart / wstart
//   k_iv_bucket_fill   block = the same frames, walked in order by one wave: pairs[] ends up sorted by
//                      (component, frame), i.e. utterance-major and in time order inside every bucket --
//                      which is what lets k_iv_stats accumulate deterministically straight from the buckets.
// (FB_IV_FB = 64 frames per partition block: fb_kernels.h)
__global__ __launch_bounds__(256) void k_iv_bucket_count(FbIvDev iv, const int *__restrict__ n_rows_ptr,
                                                         const int *__restrict__ sel, int *__restrict__ cnt) {
  extern __shared__ int s_cnt[];  // [Cpad]
  const int Cpad = iv.Cpad, nsel = iv.nsel, n_rows = *n_rows_ptr;
  for (int i = threadIdx.x; i < Cpad; i += 256) s_cnt[i] = 0;
  __syncthreads();
  const int r0 = blockIdx.x * FB_IV_FB, r1 = min(n_rows, r0 + FB_IV_FB);
  for (int e = r0 * nsel + (int)threadIdx.x; e < r1 * nsel; e += 256) {
    const int k = sel[e];
    if (k >= 0 && k < iv.C) atomicAdd(&s_cnt[k], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cpad; i += 256) cnt[(size_t)blockIdx.x * Cpad + i] = s_cnt[i];
}
// cnt[blk][k] -> pref[blk][k] = exclusive prefix over blk, hist[k] = total; then bstart[C+1] = exclusive scan of
// hist and wstart[C+1] = exclusive scan of ceil(hist / FB_IV_CH); nz[0 .. nz[C]) = the non-empty buckets.
// (a) per component: exclusive scan over the partition blocks.  Workgroup = 64 components x 16 block segments: every
//     thread first sums its segment (all loads independent), the 16 segment sums are scanned through LDS, then the
//     segment is walked again to write the prefixes -- instead of one thread walking all n_blk blocks of a component.
template <int PMAX>  // blocks per segment held in registers (0: walk the segment twice)
__global__ __launch_bounds__(1024) void k_iv_bucket_scan_blocks(int C, int Cpad, int n_blk, const int *__restrict__ cnt,
                                                                int *__restrict__ pref, int *__restrict__ hist) {
  __shared__ int ssum[16][64];
  const int kk = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kk;
  const int per = (n_blk + 15) / 16;
  const int j0 = seg * per, j1 = min(n_blk, j0 + per);
  const bool kok = k < C;
  // (a segment of at most PMAX blocks -- configs[2]'s 240 blocks are 15 per segment, configs[4]'s 942 are 59 -- is loaded ONCE, every load in flight
  //  together, and kept in registers for the second walk: the two run-time loops were eight dependent round trips)
  int v[PMAX > 0 ? PMAX : 1];
  int tot = 0;
  if (PMAX > 0) {
#pragma unroll
    for (int u = 0; u < PMAX; ++u) {
      const int j = min(j0 + u, n_blk - 1);
      const int x = kok ? cnt[(size_t)j * Cpad + k] : 0;
      v[u] = j0 + u < j1 ? x : 0;
    }
#pragma unroll
    for (int u = 0; u < PMAX; ++u) tot += v[u];
  } else if (kok) {  // (longer segments: sixteen loads in flight per trip, twice)
    for (int jc = j0; jc < j1; jc += 16) {
      int x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = cnt[(size_t)min(jc + u, n_blk - 1) * Cpad + k];
#pragma unroll
      for (int u = 0; u < 16; ++u) tot += jc + u < j1 ? x[u] : 0;
    }
  }
  ssum[seg][kk] = tot;
  __syncthreads();
  int run = 0;
  for (int q = 0; q < seg; ++q) run += ssum[q][kk];
  if (kok) {
    if (PMAX > 0) {
#pragma unroll
      for (int u = 0; u < PMAX; ++u)
        if (j0 + u < j1) {
          pref[(size_t)(j0 + u) * Cpad + k] = run;
          run += v[u];
        }
    } else {
      for (int jc = j0; jc < j1; jc += 16) {
        int x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = cnt[(size_t)min(jc + u, n_blk - 1) * Cpad + k];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (jc + u < j1) {
            pref[(size_t)(jc + u) * Cpad + k] = run;
            run += x[u];
          }
      }
    }
    if (seg == 15) hist[k] = run;  // (empty trailing segments included: run = the component's total)
  }
}
// (b) bstart[C+1] = exclusive scan of hist, wstart[C+1] = exclusive scan of ceil(hist / FB_IV_CH), nz[] = the non-empty
//     buckets.  Single workgroup.
__global__ __launch_bounds__(1024) void k_iv_bucket_scan(int C, const int *__restrict__ hist, int *__restrict__ bstart,
                                                         int *__restrict__ wstart, int *__restrict__ nz) {
  __shared__ int sa[1024], sb[1024], sc[1024];
  const int per = (C + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(C, lo + per);
  int a = 0, bsum = 0, nzc = 0;
  for (int k = lo; k < hi; ++k) { a += hist[k]; bsum += (hist[k] + FB_IV_CH - 1) / FB_IV_CH; nzc += hist[k] > 0; }
  // exclusive scan of the 1024 partials: wave scans by shuffles + 16 wave totals
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int ia = a, ib = bsum, ic = nzc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64), uc = __shfl_up(ic, o, 64);
    if (lane >= o) { ia += ua; ib += ub; ic += uc; }
  }
  if (lane == 63) { sa[w] = ia; sb[w] = ib; sc[w] = ic; }
  __syncthreads();
  int ra = ia - a, rb = ib - bsum, rc = ic - nzc;
  for (int i = 0; i < w; ++i) { ra += sa[i]; rb += sb[i]; rc += sc[i]; }
  if (threadIdx.x == 1023) { bstart[C] = ra + a; wstart[C] = rb + bsum; nz[C] = rc + nzc; }
  for (int k = lo; k < hi; ++k) {
    bstart[k] = ra;
    wstart[k] = rb;
    if (hist[k] > 0) nz[rc++] = k;  // ascending list of the non-empty buckets (-> k_iv_stats work items)
    ra += hist[k];
    rb += (hist[k] + FB_IV_CH - 1) / FB_IV_CH;
  }
}
// one wave per partition block; frames in order, the slots of a frame in parallel (their components are
// distinct, so the LDS counters are conflict-free)
__global__ __launch_bounds__(64) void k_iv_bucket_fill(FbIvDev iv, const int *__restrict__ n_rows_ptr,
                                                       const int *__restrict__ sel, const int *__restrict__ cnt,
                                                       const int *__restrict__ bstart, int *__restrict__ pairs) {
  extern __shared__ int s_pos[];  // [Cpad] next free slot of every bucket for this block
  const int Cpad = iv.Cpad, nsel = iv.nsel, n_rows = *n_rows_ptr, lane = threadIdx.x;
  const int r0 = blockIdx.x * FB_IV_FB, r1 = min(n_rows, r0 + FB_IV_FB);
  if (r0 >= r1) return;
  for (int i = lane; i < iv.C; i += 64) s_pos[i] = bstart[i] + cnt[(size_t)blockIdx.x * Cpad + i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int r = r0; r < r1; ++r) {
    if (lane < nsel) {
      const int e = r * nsel + lane;
      const int k = sel[e];
      if (k >= 0 && k < iv.C) {
        const int p = s_pos[k];
        s_pos[k] = p + 1;
        pairs[p] = e;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// (Round 5 built the four launches above as ONE -- count, the two scans and the fill separated by two grid barriers on an
// arrival counter, every exchanged word an agent-scope store / load, no device-wide fence -- and measured it: 41 - 46 us
// against the 34.6 us + three launch boundaries of the separate kernels.  On the eight-XCD MI355X a grid barrier costs
// what a kernel boundary costs -- the write-through of the block's stores, the arrival, the poll: 4 - 5 us -- and the
// phases between them are memory round trips either way.  Removed; what stayed is its fill, below.)
//
// k_iv_bucket_fill4 (round 5): the fill with FOUR waves per partition block.  The serial part of the fill is one LDS
// read-modify-write per frame (a frame's slots in parallel: their components are distinct); a block of 64 frames walked by
// one wave was 16 us.  Here the block first counts its four 16-frame sub-blocks apart (LDS histograms, as
// k_iv_bucket_count does for the whole block), turns them into the sub-blocks' first slots -- bucket start + the blocks
// before this one (pref) + the sub-blocks before this one --, and every wave walks its own 16 frames: the partition is the
// same stable one (sub-blocks in order, frames in order).
__global__ __launch_bounds__(256) void k_iv_bucket_fill4(FbIvDev iv, const int *__restrict__ n_rows_ptr,
                                                         const int *__restrict__ sel, const int *__restrict__ pref,
                                                         const int *__restrict__ bstart, int *__restrict__ pairs) {
  extern __shared__ int s_h[];                  // [4][Cpad]: the sub-blocks' histograms, then their first slots
  const int C = iv.C, Cpad = iv.Cpad, nsel = iv.nsel, n_rows = *n_rows_ptr;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, blk = blockIdx.x;
  const int r0 = blk * FB_IV_FB, r1 = min(n_rows, r0 + FB_IV_FB);
  if (r0 >= r1) return;
  const int q0 = r0 + 16 * w, q1 = min(r1, q0 + 16);
  int kv[16];   // the 16 frames' selections of this slot: one batch of loads for both passes
#pragma unroll
  for (int u = 0; u < 16; ++u) kv[u] = (lane < nsel && q0 + u < q1) ? sel[(q0 + u) * nsel + lane] : -1;
  for (int i = tid; i < 4 * Cpad; i += 256) s_h[i] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (kv[u] >= 0 && kv[u] < C) atomicAdd(&s_h[w * Cpad + kv[u]], 1);
  __syncthreads();
  for (int k0 = tid; k0 < C; k0 += 8 * 256) {  // (eight components' two loads in flight per trip, not one per trip)
    int pb[8], pp[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kc = min(k0 + 256 * u, C - 1);
      pb[u] = bstart[kc];
      pp[u] = pref[(size_t)blk * Cpad + kc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 256 * u;
      if (k < C) {
        const int p0 = pb[u] + pp[u];
        const int c0 = s_h[k], c1 = s_h[Cpad + k], c2 = s_h[2 * Cpad + k];
        s_h[k] = p0; s_h[Cpad + k] = p0 + c0; s_h[2 * Cpad + k] = p0 + c0 + c1; s_h[3 * Cpad + k] = p0 + c0 + c1 + c2;
      }
    }
  }
  __syncthreads();
  int *pos = s_h + w * Cpad;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int r = q0 + u;
    if (r >= q1) break;
    const int k = kv[u];
    if (k >= 0 && k < C) {
      const int p = pos[k];
      pos[k] = p + 1;
      pairs[p] = r * nsel + lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// TRI64 = ceil(D(D+1)/2 / 64): 42 for D = 72 (the recipe), 52 covers D <= 80
template <int TRI64>
__global__ __launch_bounds__(256) void k_iv_fullcov(FbIvDev iv, const float *__restrict__ feats,
                                                    const int *__restrict__ bstart, const int *__restrict__ wstart,
                                                    const int *__restrict__ pairs, float *__restrict__ llf) {
  __shared__ float xs[4][128];
  const int n_work = wstart[iv.C];
  const int wi = blockIdx.x;
  if (wi >= n_work) return;
  // component owning work item wi: largest k with wstart[k] <= wi
  int lo = 0, hi = iv.C;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (wstart[mid] <= wi) lo = mid; else hi = mid;
  }
  const int k = lo;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = iv.D, triD = iv.triD, nsel = iv.nsel;
  const int e0 = bstart[k] + (wi - wstart[k]) * FB_IV_CH;
  const int e1 = min(bstart[k + 1], e0 + FB_IV_CH);
  // this lane's entries of the packed precision matrix, with the 1/2 of the diagonal folded in
  float pk[TRI64];
  unsigned short rc[TRI64];
  const float *P = iv.fg_P + (size_t)k * triD;
#pragma unroll
  for (int i = 0; i < TRI64; ++i) {
    const int e = lane + 64 * i;
    const int ec = min(e, triD - 1);
    const int r = iv.tri_r[ec], c = iv.tri_c[ec];
    float v = (e < triD) ? P[ec] : 0.0f;
    if (r == c) v *= 0.5f;
    pk[i] = v;
    rc[i] = (unsigned short)(r | (c << 8));
  }
  const float *mic = iv.fg_mic + (size_t)k * D;
  const double m0 = lane < D ? (double)mic[lane] : 0.0, m1 = lane + 64 < D ? (double)mic[lane + 64] : 0.0;
  const double gck = (double)iv.fg_gconsts[k];
  float *x = xs[w];
  for (int e = e0 + w; e < e1; e += 4) {
    const int pr = pairs[e];
    const int row = pr / nsel;
    for (int i = lane; i < D; i += 64) x[i] = feats[(size_t)row * D + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < TRI64; ++i) {
      const double xr = (double)x[rc[i] & 0xff], xc = (double)x[rc[i] >> 8];
      acc = fma(-(double)pk[i], xr * xc, acc);
    }
    if (lane < D) acc = fma(m0, (double)x[lane], acc);
    if (lane + 64 < D) acc = fma(m1, (double)x[lane + 64], acc);
    acc = fb_wave_sum(acc);
    if (lane == 0) llf[pr] = (float)(gck + acc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Same numbers, D known at compile time: thread = (frame, slot) pair of the component's bucket with the
// frame in registers (float64); the component's float64 record (packed precision matrix with halved
// diagonal, linear term, gconst: fg64) is staged in LDS by one coalesced copy per workgroup and read back
// as broadcasts -- no cross-lane reduction.  1/2 x'Px = sum_r x_r (sum_{c<r} P_rc x_c + 1/2 P_rr x_r):
// D(D+1)/2 + D fma per pair; bound by the LDS issue rate.  (A version feeding the matrix through scalar
// loads / SGPR operands waited on an L2 round trip every 16 entries -- the 21 KB record does not fit the
// scalar cache -- and was 1.7x slower.)
template <int D>
__global__ __launch_bounds__(FB_IV_CH) void k_iv_fullcov_lds(FbIvDev iv, const float *__restrict__ feats,
                                                             const int *__restrict__ bstart,
                                                             const int *__restrict__ wstart,
                                                             const int *__restrict__ pairs, float *__restrict__ llf) {
  constexpr int TRI = D * (D + 1) / 2, REC = TRI + D + 1;
  __shared__ __attribute__((aligned(16))) double sP[(REC + 1) & ~1];
  const int n_work = wstart[iv.C];
  const int wi = blockIdx.x;
  if (wi >= n_work) return;
  int lo = 0, hi = iv.C;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (wstart[mid] <= wi) lo = mid; else hi = mid;
  }
  const int k = lo;
  {
    const double *src = iv.fg64 + (size_t)k * REC;
    for (int i = threadIdx.x; i < REC; i += FB_IV_CH) sP[i] = src[i];
  }
  const int e0 = bstart[k] + (wi - wstart[k]) * FB_IV_CH;
  const int e1 = min(bstart[k + 1], e0 + FB_IV_CH);
  const int e = e0 + (int)threadIdx.x;
  const bool ok = e < e1;
  const int pr = pairs[ok ? e : e0];
  const int row = pr / iv.nsel;
  double x[D];
  {
    const float4 *fr = reinterpret_cast<const float4 *>(feats + (size_t)row * D);  // D % 4 == 0
#pragma unroll
    for (int q = 0; q < D / 4; ++q) {
      const float4 v = fr[q];
      x[4 * q] = (double)v.x; x[4 * q + 1] = (double)v.y; x[4 * q + 2] = (double)v.z; x[4 * q + 3] = (double)v.w;
    }
  }
  __syncthreads();
  // chunks of 16 entries: the next chunk is read while the current one is consumed; an empty asm that the
  // accumulators pass through (with a memory clobber) keeps the compiler from hoisting the whole record
  // into registers
  double cur[16], nxt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { cur[i] = sP[i]; nxt[i] = sP[16 + i]; }
  double half_quad = 0.0;
#pragma unroll
  for (int r = 0; r < D; ++r) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      const int idx = r * (r + 1) / 2 + c;  // compile-time after unrolling
      if (idx > 0 && (idx & 15) == 0) {
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(half_quad) : : "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
        if (idx + 16 < TRI) {
#pragma unroll
          for (int i = 0; i < 16; ++i) nxt[i] = sP[idx + 16 + i];  // (the record is longer than TRI: stays in bounds)
        }
      }
      const double pv = cur[idx & 15];
      if (c & 1) a1 = fma(pv, x[c], a1); else a0 = fma(pv, x[c], a0);
    }
    half_quad = fma(x[r], a0 + a1, half_quad);
  }
  double lin = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) lin = fma(sP[TRI + d], x[d], lin);
  if (ok) llf[pr] = (float)(sP[TRI + D] + (lin - half_quad));
}

// The same log-likelihoods on the float64 matrix cores (round 4).  With P = L L' (Cholesky, fb_load_ivector) and
// mu = P^-1 means_invcovars the record's function gconst + means_invcovars . x - 1/2 x'Px is
//     gc'' - 1/2 |L'(x - mu)|^2,        gc'' = gconst + 1/2 mu'P mu
// (an identity for any mu up to the term (means_invcovars - P mu) . x, which the host drives below 1e-18 by refining
// mu in long double), so a bucket's pairs are ONE product Y = (X - 1 mu') L with the lower-triangular L, followed by
// row sums of squares: v_mfma_f64_16x16x4_f64 tiles of 16 pairs x 16 columns, K = 4 rows of L per instruction, and
// only the K steps at or below the tile's diagonal -- 18 + 14 + 10 + 6 + 2 = 50 instructions per 16 pairs (102 kflop
// against the 2 700 fma = 5.4 kflop per pair of the triangle form: 1.2 x the arithmetic, none of its 1 300 LDS broadcasts
// per wave).  Workgroup = (component, chunk of 256 pairs) as before, wave = 64 pairs = four row tiles that share every
// fragment of L (read once per wave from the record staged in LDS); a lane holds element (pair l & 15, K place l >> 4)
// of its tiles' A operands -- its pair's features minus mu, 18 values per tile, gathered straight from the feature rows
// -- and 4 results per tile, squared and added as the column tiles finish; the 16 lanes of a K place are summed at the
// end.  Record (fgL, per component): 50 fragments x 64 lanes (fragment (jt, s), s >= 4 jt: lane -> L[row(s, l >> 4)][16 jt +
// (l & 15)] with the kernel's K order row(s, q) = 16 (s / 4) + 4 q + s % 4 (64 + 2 q + s - 16 for s >= 16), zero above the
// diagonal and past D), then mu[72], then gc''.
#define FB_FCM_NFRAG 50
#define FB_FCM_REC (FB_FCM_NFRAG * 64 + 72 + 2)
// A wave takes NRT row tiles of 16 pairs (the workgroup's FB_IV_CH pairs over FB_IV_CH / (16 NRT) waves): NRT = 1 keeps
// the wave under 128 registers -- four or more waves per SIMD, which is what hides the record's copy, the pair -> feature
// row round trips and the matrix pipe's latency behind one another (four tiles per wave, 340 registers, one wave per
// SIMD: 116 us against the triangle form's 95; two tiles: 69; one: 60.5 -- measured).
__device__ __forceinline__ int fb_wave_lower_bound(const int *__restrict__ a, int lo, int hi, int key, int lane);
template <int NRT>
__global__ __launch_bounds__(FB_IV_CH * 4 / NRT) __attribute__((amdgpu_waves_per_eu(NRT == 1 ? 8 : 3, 8))) void k_iv_fullcov_mfma(FbIvDev iv, const float *__restrict__ feats,
                                                                       const int *__restrict__ bstart,
                                                                       const int *__restrict__ wstart,
                                                                       const int *__restrict__ pairs, float *__restrict__ llf) {
  constexpr int D = 72, NS = D / 4, NT = FB_IV_CH * 4 / NRT;
  __shared__ __attribute__((aligned(16))) double sL[FB_FCM_REC];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = blockIdx.x;
  if (wi >= wstart[iv.C]) return;
  // the component of this work item: the last k with wstart[k] <= wi (two rounds of 64 probes; empty buckets repeat
  // their successor's start and are skipped by "last")
  const int k = __builtin_amdgcn_readfirstlane(fb_wave_lower_bound(wstart, 0, iv.C + 1, wi + 1, lane)) - 1;
  const int b0 = __builtin_amdgcn_readfirstlane(bstart[k]), b1 = __builtin_amdgcn_readfirstlane(bstart[k + 1]);
  const int c0 = b0 + (wi - __builtin_amdgcn_readfirstlane(wstart[k])) * FB_IV_CH;  // the chunk's first pair
  const int e0 = c0 + 16 * NRT * w, e1 = min(b1, c0 + FB_IV_CH);
  const bool any = e0 < e1;  // (a scalar)
  // the pairs' feature values first (two dependent global round trips), the record's copy behind them
  const int li = lane & 15, kq = lane >> 4;
  int pr[NRT];
  float xf[NRT][NS];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    const int p = e0 + 16 * rt + li;
    pr[rt] = any ? pairs[p < e1 ? p : e0] : 0;
  }
  // K place (step s2, l >> 4) = dimension 16 (s2 / 4) + 4 (l >> 4) + s2 % 4 (64 + 2 (l >> 4) + s2 - 16 for the last two steps):
  // the order within a 16-row block of L is free, and this one makes a lane's values of a block ONE 16-byte piece of its
  // pair's feature row (five loads per tile instead of eighteen, whole 64-byte sectors per row)
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    const float *fr = feats + (size_t)(pr[rt] / iv.nsel) * D;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float4 v = any ? *reinterpret_cast<const float4 *>(fr + 16 * b + 4 * kq) : float4{0.f, 0.f, 0.f, 0.f};
      xf[rt][4 * b] = v.x; xf[rt][4 * b + 1] = v.y; xf[rt][4 * b + 2] = v.z; xf[rt][4 * b + 3] = v.w;
    }
    const float2 v2 = any ? *reinterpret_cast<const float2 *>(fr + 64 + 2 * kq) : float2{0.f, 0.f};
    xf[rt][16] = v2.x; xf[rt][17] = v2.y;
  }
  {
    const double2 *src = reinterpret_cast<const double2 *>(iv.fgL + (size_t)k * FB_FCM_REC);
    double2 *dst = reinterpret_cast<double2 *>(sL);
    constexpr int NCP = (FB_FCM_REC / 2 + NT - 1) / NT;  // every load of the copy in flight before the first store
    double2 tmp[NCP];
#pragma unroll
    for (int j = 0; j < NCP; ++j) {
      const int i = (int)threadIdx.x + j * NT;
      tmp[j] = src[i < FB_FCM_REC / 2 ? i : 0];
    }
#pragma unroll
    for (int j = 0; j < NCP; ++j) {
      const int i = (int)threadIdx.x + j * NT;
      if (i < FB_FCM_REC / 2) dst[i] = tmp[j];
    }
  }
  __syncthreads();
  if (!any) return;
  const double *mu = sL + FB_FCM_NFRAG * 64;
  double a[NRT][NS];
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2) {
    const double m = mu[s2 < 16 ? 16 * (s2 / 4) + 4 * kq + (s2 % 4) : 64 + 2 * kq + (s2 - 16)];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) a[rt][s2] = (double)xf[rt][s2] - m;
  }
  double acc[NRT][4];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[rt][r] = 0.0;
  int f = 0;
#pragma unroll
  for (int jt = 0; jt < 5; ++jt) {
    fb_d4 c[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) c[rt] = fb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s2 = 4 * jt; s2 < NS; ++s2, ++f) {
      const double b = sL[f * 64 + lane];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) c[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt][s2], b, c[rt], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rt][r] = fma(c[rt][r], c[rt][r], acc[rt][r]);
  }
  const double gc2 = sL[FB_FCM_NFRAG * 64 + D];
  // a result register holds (pair (l >> 4) + 4 r of the tile, column l & 15): sum the 16 columns of a K-place group
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = acc[rt][r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int i = kq + 4 * r;                       // the pair of the tile this lane group's register r belongs to
      const int pri = __shfl(pr[rt], i, 64);          // (lane i, K place 0, holds that pair's index)
      if (li == 0 && e0 + 16 * rt + i < e1) llf[pri] = (float)(gc2 - 0.5 * v);
    }
}

// softmax over the nsel full-covariance log-likelihoods of a frame + min-post pruning: lane = slot
__global__ __launch_bounds__(256) void k_iv_post(FbIvDev iv, const int *__restrict__ n_rows_ptr,
                                                 const int *__restrict__ sel, const float *__restrict__ llf,
                                                 float *__restrict__ post) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= *n_rows_ptr) return;
  const int nsel = iv.nsel;
  int k = -1;
  double my_ll = -INFINITY;
  if (lane < nsel) {
    k = sel[(size_t)row * nsel + lane];
    if (k >= 0 && k < iv.C) my_ll = (double)llf[(size_t)row * nsel + lane];
  }
  const double mx = fb_wave_max(my_ll);
  const double ex = (my_ll > -INFINITY) ? exp(my_ll - mx) : 0.0;
  const double sum = fb_wave_sum(ex);
  float p = (lane < nsel) ? (float)(ex / sum) : 0.0f;
  const float min_post = iv.min_post;
  if (min_post != 0.0f) {
    // first slot holding the maximum posterior (Vector::Max(&index) semantics: first max)
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
  if (lane < nsel) post[(size_t)row * nsel + lane] = p;
}

// The same with TWO frames per wave (num_gselect <= 32: the recipe's 20): a frame's slots are the lanes of one half,
// every reduction is the lower five steps of the 64-lane butterflies above -- whose first step only adds the zeros /
// compares the -inf of the unused upper half --, so the posteriors are the same bit for bit with half the instructions
// (the kernel is bound by its float64 exp / division / shuffles per wave: 11.0 -> ~6 us at 15 300 frames).
__global__ __launch_bounds__(256) void k_iv_post2(FbIvDev iv, const int *__restrict__ n_rows_ptr,
                                                  const int *__restrict__ sel, const float *__restrict__ llf,
                                                  float *__restrict__ post) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, slot = lane & 31;
  const int row = blockIdx.x * 8 + w * 2 + (lane >> 5);
  const int n_rows = *n_rows_ptr;
  if (blockIdx.x * 8 + w * 2 >= n_rows) return;      // (wave-uniform: both frames of the wave are past the end)
  const bool rok = row < n_rows;
  const int nsel = iv.nsel;
  int k = -1;
  double my_ll = -INFINITY;
  if (rok && slot < nsel) {
    k = sel[(size_t)row * nsel + slot];
    if (k >= 0 && k < iv.C) my_ll = (double)llf[(size_t)row * nsel + slot];
  }
  double mx = my_ll;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { const double v = __shfl_xor(mx, o, 64); mx = v > mx ? v : mx; }
  const double ex = (my_ll > -INFINITY) ? exp(my_ll - mx) : 0.0;
  double sum = ex;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  float p = (slot < nsel) ? (float)(ex / sum) : 0.0f;
  const float min_post = iv.min_post;
  if (min_post != 0.0f) {
    float pm = p;
    int pi = slot < nsel ? slot : 64;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float ov = __shfl_xor(pm, o, 64);
      const int oi = __shfl_xor(pi, o, 64);
      if (ov > pm || (ov == pm && oi < pi)) { pm = ov; pi = oi; }
    }
    if (p < min_post) p = 0.0f;
    double s2 = (double)p;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    if (s2 == 0.0) p = (slot == pi) ? 1.0f : 0.0f;
    else p = (float)((double)p / s2);
  }
  if (rok && slot < nsel) post[(size_t)row * nsel + slot] = p;
}

// bucket_ws (ints): hist[C], bstart[C+1], wstart[C+1], flags[C] (zero on entry, see k_iv_active), nz[C+1],
// cnt[n_blk][Cpad], pref[n_blk][Cpad] with n_blk = ceil(rows_cap / FB_IV_FB)
size_t fb_iv_bucket_ws_ints(const FbIvDev &iv, int rows_cap) {
  return (size_t)5 * iv.C + 3 + (size_t)2 * ((rows_cap + FB_IV_FB - 1) / FB_IV_FB) * iv.Cpad;
}
int *fb_iv_bucket_cnt(const FbIvDev &iv, int *bucket_ws) {   // (the layout fb_launch_iv_select_post carves out below)
  const int C = iv.C;
  return bucket_ws + C + (C + 1) + (C + 1) + C + (C + 1);
}
void fb_launch_iv_select_post(hipStream_t s, const FbIvDev &iv, const float *ll, const float *feats,
                              const int *n_rows_ptr, int rows_cap, int *sel, float *post, int *bucket_ws,
                              int *pairs, float *llf, const int *sel_gate, bool run_select, bool run_count) {
  if (rows_cap <= 0) return;
  const int C = iv.C;
  int *hist = bucket_ws, *bstart = hist + C, *wstart = bstart + (C + 1), *nz = wstart + (C + 1) + C,
      *cnt = nz + (C + 1);
  const int n_blk = (rows_cap + FB_IV_FB - 1) / FB_IV_FB;
  int *pref = cnt + (size_t)n_blk * iv.Cpad;
  if (run_select) {   // (false: sel[] already holds the selection -- fb_launch_gsel_wide)
    const dim3 grid((rows_cap + 3) / 4), blk(256);
    const int nj = (iv.Cpad + 63) / 64;
    if (nj <= 4) hipLaunchKernelGGL(k_iv_select<4>, grid, blk, 0, s, iv, ll, n_rows_ptr, sel, sel_gate);
    else if (nj <= 8) hipLaunchKernelGGL(k_iv_select<8>, grid, blk, 0, s, iv, ll, n_rows_ptr, sel, sel_gate);
    else if (nj <= 16) hipLaunchKernelGGL(k_iv_select<16>, grid, blk, 0, s, iv, ll, n_rows_ptr, sel, sel_gate);
    else if (nj <= 32) hipLaunchKernelGGL(k_iv_select<32>, grid, blk, 0, s, iv, ll, n_rows_ptr, sel, sel_gate);
    else hipLaunchKernelGGL(k_iv_select<64>, grid, blk, 0, s, iv, ll, n_rows_ptr, sel, sel_gate);  // C <= 4096 (fb_load_ivector)
  }
  const size_t lds_c = sizeof(int) * (size_t)iv.Cpad;
  if (run_count)   // (false: k_gsel_final_w has made the counts -- fb_launch_gsel_wide)
    hipLaunchKernelGGL(k_iv_bucket_count, dim3(n_blk), dim3(256), lds_c, s, iv, n_rows_ptr, sel, cnt);
  {
    const int per = (n_blk + 15) / 16;
    const dim3 grid((C + 63) / 64), blk(1024);
    if (per <= 16) hipLaunchKernelGGL(k_iv_bucket_scan_blocks<16>, grid, blk, 0, s, C, iv.Cpad, n_blk, cnt, pref, hist);
    else hipLaunchKernelGGL(k_iv_bucket_scan_blocks<0>, grid, blk, 0, s, C, iv.Cpad, n_blk, cnt, pref, hist);
  }
  hipLaunchKernelGGL(k_iv_bucket_scan, dim3(1), dim3(1024), 0, s, C, hist, bstart, wstart, nz);
  if (4 * lds_c <= 64 * 1024 && getenv("FB_IV_FILL1") == nullptr)
    hipLaunchKernelGGL(k_iv_bucket_fill4, dim3(n_blk), dim3(256), 4 * lds_c, s, iv, n_rows_ptr, sel, pref, bstart, pairs);
  else   // (FB_IV_FILL1=1: the one-wave fill, A/B and tests)
    hipLaunchKernelGGL(k_iv_bucket_fill, dim3(n_blk), dim3(64), lds_c, s, iv, n_rows_ptr, sel, pref, bstart, pairs);
  const int n_pairs_cap = rows_cap * iv.nsel;
  const int work_cap = C + (n_pairs_cap + FB_IV_CH - 1) / FB_IV_CH;
  const char *fc_env = getenv("FB_IV_FULLCOV");  // "lds": the triangle-form kernel (A/B runs, tests); read per launch
  const bool fc_lds = fc_env && strcmp(fc_env, "lds") == 0;
  if (iv.D == 72 && iv.fgL && !fc_lds)
    hipLaunchKernelGGL(k_iv_fullcov_mfma<1>, dim3(work_cap), dim3(FB_IV_CH * 4), 0, s, iv, feats, bstart, wstart, pairs, llf);
  else if (iv.D == 72)
    hipLaunchKernelGGL((k_iv_fullcov_lds<72>), dim3(work_cap), dim3(FB_IV_CH), 0, s, iv, feats, bstart, wstart, pairs, llf);
  else if (iv.triD <= 42 * 64)
    hipLaunchKernelGGL((k_iv_fullcov<42>), dim3(work_cap), dim3(256), 0, s, iv, feats, bstart, wstart, pairs, llf);
  else
    hipLaunchKernelGGL((k_iv_fullcov<52>), dim3(work_cap), dim3(256), 0, s, iv, feats, bstart, wstart, pairs, llf);
  if (iv.nsel <= 32) hipLaunchKernelGGL(k_iv_post2, dim3((rows_cap + 7) / 8), dim3(256), 0, s, iv, n_rows_ptr, sel, llf, post);
  else hipLaunchKernelGGL(k_iv_post, dim3((rows_cap + 3) / 4), dim3(256), 0, s, iv, n_rows_ptr, sel, llf, post);
}

// ------------------------------------------------------- statistics (K10a)
// gamma[b][k] = sum_t post, X[b][k][:] = sum_t post * x_t, accumulated straight from the component's bucket:
// the stable partition left its (frame, slot) pairs in (utterance, time) order, so the pairs of one
// (component, utterance) are a contiguous run.  FB_IV_SSPLIT workgroups per component, wave = utterance
// (64 in flight per component): the wave adds the frames of its run in order -- lane d owns
// dimension d (and d + 64), float64, the same operations in the same order as a frame-major loop -- with
// the feature rows of the next 16 pairs in flight.  Only the ~15 % of the Gaussians that were selected at all cost
// anything; utterances without a pair of this component get explicit zeros (the contraction reads whole
// rows of an active component).  flags[k] = 1 when any posterior is non-zero (-> k_iv_active).
__device__ __forceinline__ int fb_wave_lower_bound(const int *__restrict__ a, int lo, int hi, int key, int lane) {
  // first index in [lo, hi) with a[idx] >= key (hi if none); a[] ascending; all lanes return the same value.
  // 64 probes per round instead of one: a range of n elements needs ceil(log64 n) dependent loads.
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) / 64;
    const int idx = lo + lane * step;
    const int v = idx < hi ? a[idx] : 0x7fffffff;
    const unsigned long long ge = __ballot(v >= key);
    const int first = ge ? (int)__builtin_ctzll(ge) : 64;  // probes below `first` are < key
    if (first == 0) return lo;
    const int nhi = first == 64 ? hi : min(hi, lo + first * step);
    lo = lo + (first - 1) * step + 1;
    hi = nhi;
  }
  const int idx = lo + lane;
  const int v = idx < hi ? a[idx] : 0x7fffffff;
  const unsigned long long ge = __ballot(v >= key);
  return ge ? lo + (int)__builtin_ctzll(ge) : hi;
}
// (Round 5, measured and removed: the bucket's pairs staged in LDS for the searches and reads of a run -- 62 -> 80 us, the
//  copy + barrier per item and the occupancy the 16 KB cost outweigh the dependent round trips the other waves hide --; the
//  active list made by this kernel's last workgroup instead of k_iv_active's own launch -- 62.3 + 4.4 -> 73 us: 2 048
//  arrivals on one counter cost more than the launch; every item's bucket bounds fetched up front by lane j, the run bounds
//  of a bucket of <= 64 pairs from ONE load of its pairs (popcount of entries below the key, the run's pairs shuffled out
//  of those registers), the next item's pairs requested ahead -- 7 dependent round trips per item down to 2, and 62 -> 71 us
//  (spd 200: 268 -> 321): at eight resident waves per SIMD the launch is bound by instructions issued per item, not by
//  the chain, and the shuffles and the 78 registers (six waves) cost more than the round trips the other waves hide.)
#define FB_IV_SG 8  // feature rows in flight per wave (two register buffers of this size).  Round 4: 16 -> 8 -- the kernel is a
                    // chain of dependent round trips per (bucket, utterance) run, hidden by OTHER waves: 50 registers and eight waves
                    // per SIMD (the whole grid resident) beat the deeper prefetch at 100 registers, 84 -> 61 us (spd 200: 354 -> 267);
                    // 4 measures the same, 32 spills
#define FB_IV_SSPLIT 16 // workgroups per component (utterances dealt round-robin): spreads the few very popular
                        // components, whose buckets hold thousands of pairs, over several CUs
__global__ __launch_bounds__(256) void k_iv_stats(FbIvDev iv, const float *__restrict__ feats,
                                                   const int *__restrict__ row_off, const int *__restrict__ pairs,
                                                   const int *__restrict__ bstart, const int *__restrict__ nz,
                                                   const float *__restrict__ post, int B, int Bpad,
                                                   double *__restrict__ gammaT, double *__restrict__ XT,
                                                   int *__restrict__ flags) {
  const int D = iv.D, nsel = iv.nsel;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // results of the 4 waves (4 consecutive utterances) are gathered here and leave as 32-byte row segments: single
  // 8-byte stores scattered Bpad*8 bytes apart cost 12x their size in HBM write traffic (PMC: 119 MB for 10 MB)
  __shared__ __attribute__((aligned(32))) double s_out[129 * 4];
  const int n_items = nz[iv.C] * FB_IV_SSPLIT;  // work item = (non-empty bucket, utterance residue class)
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
  const int k = nz[item / FB_IV_SSPLIT], ys = item % FB_IV_SSPLIT;
  const int e0 = bstart[k], e1 = bstart[k + 1];
  const bool two = lane + 64 < D;  // this lane also owns dimension lane + 64 (D <= 128)
  bool any = false;
  for (int b0 = 4 * ys; b0 < B; b0 += 4 * FB_IV_SSPLIT) {
    const int b = b0 + w;
    double acc0 = 0.0, acc1 = 0.0, gam = 0.0;
    if (b < B) {
    // the run of this utterance inside the bucket: two 64-ary searches (<= 3 dependent loads each)
    const int s0 = fb_wave_lower_bound(pairs, e0, e1, row_off[b] * nsel, lane);
    const int s1 = fb_wave_lower_bound(pairs, s0, e1, row_off[b + 1] * nsel, lane);
    for (int c0 = s0; c0 < s1; c0 += 64) {
      const int n = min(64, s1 - c0);
      int pr = 0;
      float p = 0.0f;
      if (lane < n) { pr = pairs[c0 + lane]; p = post[pr]; }
      const int rowl = pr / nsel;  // (lanes >= n: p = 0, row 0 -- loads stay in bounds, contributions are skipped)
      float xa0[FB_IV_SG], xa1[FB_IV_SG], xb0[FB_IV_SG], xb1[FB_IV_SG];
      auto fetch = [&](int i0, float (&x0)[FB_IV_SG], float (&x1)[FB_IV_SG]) {
#pragma unroll
        for (int u = 0; u < FB_IV_SG; ++u) {
          const int row = __shfl(rowl, min(i0 + u, 63), 64);
          const float *fr = feats + (size_t)row * D;
          x0[u] = lane < D ? fr[lane] : 0.0f;
          x1[u] = two ? fr[lane + 64] : 0.0f;
        }
      };
      auto add = [&](int i0, const float (&x0)[FB_IV_SG], const float (&x1)[FB_IV_SG]) {
#pragma unroll
        for (int u = 0; u < FB_IV_SG; ++u) {
          const float pv = __shfl(p, min(i0 + u, 63), 64);
          if (i0 + u < n && pv != 0.0f) {  // wave-uniform
            const double wv = (double)pv;
            gam = __dadd_rn(gam, wv);
            acc0 = __dadd_rn(acc0, __dmul_rn(wv, (double)x0[u]));
            acc1 = __dadd_rn(acc1, __dmul_rn(wv, (double)x1[u]));
            any = true;
          }
        }
      };
      // frames strictly in order; the rows of the next group are requested before the current one is added
      fetch(0, xa0, xa1);
      for (int i0 = 0; i0 < n; i0 += 2 * FB_IV_SG) {
        if (i0 + FB_IV_SG < n) fetch(i0 + FB_IV_SG, xb0, xb1);
        add(i0, xa0, xa1);
        if (i0 + FB_IV_SG < n) {
          if (i0 + 2 * FB_IV_SG < n) fetch(i0 + 2 * FB_IV_SG, xa0, xa1);
          add(i0 + FB_IV_SG, xb0, xb1);
        }
      }
    }
    }
    if (lane < D) s_out[lane * 4 + w] = acc0;
    if (two) s_out[(lane + 64) * 4 + w] = acc1;
    if (lane == 0) s_out[128 * 4 + w] = gam;
    __syncthreads();
    {  // b0 is a multiple of 4 and Bpad a multiple of 32: the 4 columns exist (columns >= B receive zeros)
      const int t = threadIdx.x;
      if (t < D) {
        const double4 v = *reinterpret_cast<const double4 *>(&s_out[t * 4]);
        *reinterpret_cast<double4 *>(&XT[((size_t)k * D + t) * Bpad + b0]) = v;
      } else if (t == 128) {
        const double4 v = *reinterpret_cast<const double4 *>(&s_out[128 * 4]);
        *reinterpret_cast<double4 *>(&gammaT[(size_t)k * Bpad + b0]) = v;
      }
    }
    __syncthreads();
  }
  if (lane == 0 && any) flags[k] = 1;
  }
}
void fb_launch_iv_stats(hipStream_t s, const FbIvDev &iv, const float *feats, const int *row_off, const int *pairs,
                        const int *bucket_ws, const float *post, int B, int Bpad, double *gammaT, double *XT) {
  const int *bstart = bucket_ws + iv.C;
  int *flags = const_cast<int *>(bucket_ws) + 3 * (size_t)iv.C + 2;
  const int *nz = bucket_ws + 4 * (size_t)iv.C + 2;
  hipLaunchKernelGGL(k_iv_stats, dim3(2048), dim3(256), 0, s, iv, feats, row_off, pairs, bstart, nz, post, B, Bpad,
                     gammaT, XT, flags);
}

// ---------------------------------------------- T-matrix contraction (K10b)
// Components that no utterance of the batch gave any posterior mass contribute nothing to lin / quad
// (Kaldi's GetIvectorDistMean skips gamma == 0 as well).  After gselect(20) + min-post pruning only a
// few hundred of the C Gaussians are touched by a 3-s utterance, and the 51 utterances of an NES batch
// are noisy copies of one utterance, so the contraction streams only the active rows of Sigma^-1 M / U.
__global__ __launch_bounds__(1024) void k_iv_active(int C, int *__restrict__ flags, int *__restrict__ active,
                                                    int *__restrict__ n_active, int *__restrict__ fail) {
  __shared__ int s_cnt[17];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k0 = 0; k0 < C; k0 += 1024) {  // ascending k: deterministic list order
    const int k = k0 + threadIdx.x;
    int any = 0;
    if (k < C) { any = flags[k]; flags[k] = 0; }  // (k_iv_stats raises them; left clean for the next batch)
    const unsigned long long bal = __ballot(any);
    if (lane == 0) s_cnt[w] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int i = 0; i < w; ++i) off += s_cnt[i];
    if (any) active[off + __popcll(bal & ((1ull << lane) - 1ull))] = k;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int i = 0; i < 16; ++i) t += s_cnt[i]; s_base += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *n_active = s_base; *fail = 0; }  // (the solve's not-positive-definite flag: clean for this batch -- a
                                                            // hipMemsetAsync per batch was 5 us of the stream's time)
}

// The two contractions as one LDS-tiled GEMM on the float64 matrix cores (v_mfma_f64_16x16x4_f64):
//   out[b][n] = sum_q AT[q][b] * P[q][n],   q = the rows of the ACTIVE components only,
// P = U (quad: one row per component, n = packed element) or Sigma^-1 M (lin: D rows per component,
// n = i-vector dimension, split over blockIdx.y chunks of the active list whose partials the solve
// kernel adds up).  Workgroup tile = 64 utterances x 128 columns, K in stages of FB_CG_KC parameter rows:
// every row segment is 1 KB of contiguous HBM, fetched once with 16-byte loads into registers while the
// previous stage is multiplied, then parked in LDS (row strides padded so that the two half-waves of a
// fragment read hit disjoint banks).  Wave = 32 columns x 64 utterances = 8 accumulator tiles.
// The f64 MFMA peak equals the vector peak (78.6 TF); what it buys is 2048 flops per instruction and two
// 8-byte operand reads per lane instead of one per fma, which is what lets the parameter stream run at
// HBM speed.
#define FB_CG_NT 128   // columns per workgroup
#define FB_CG_LDA 80   // doubles per staged coefficient row (64 + 16: bank offset 32 between rows)
#define FB_CG_LDB 144  // doubles per staged parameter row (128 + 16)
template <bool IS_LIN, int FB_CG_KC>  // FB_CG_KC = parameter rows per stage (measured best: 16 for lin, 8 for quad)
__global__ __launch_bounds__(256) void k_iv_contract_gemm(FbIvDev iv, const double *__restrict__ AT, int ldA,
                                                          const int *__restrict__ active,
                                                          const int *__restrict__ n_active, int B,
                                                          int n_kchunks, double *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) double sA[2][FB_CG_KC * FB_CG_LDA];
  __shared__ __attribute__((aligned(16))) double sB[2][FB_CG_KC * FB_CG_LDB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int N = IS_LIN ? iv.R : iv.triR;
  const int DK = IS_LIN ? iv.D : 1;
  const double *P = IS_LIN ? iv.sim : iv.u;
  const int n0 = blockIdx.x * FB_CG_NT;
  const int b0 = blockIdx.z * 64;
  const int na = *n_active;
  int a0 = 0, a1 = na;
  if (IS_LIN) {
    const int per = (na + n_kchunks - 1) / n_kchunks;
    a0 = blockIdx.y * per;
    a1 = min(na, a0 + per);
  }
  const int nq = max(0, a1 - a0) * DK;
  const int nst = (nq + FB_CG_KC - 1) / FB_CG_KC;
  // loader roles: a parameter stage is KC rows x 64 16-byte pieces, a coefficient stage KC rows x 32
  constexpr int NPB = FB_CG_KC * 64 / 256;           // 16-byte parameter pieces per thread and stage
  constexpr int NPA = FB_CG_KC * 32 / 256;           // coefficient pieces per thread and stage
  double2 rb[NPB], ra[NPA];
  auto row_of = [&](int q) -> size_t {                // parameter / coefficient row of contraction index q
    const int qc = min(q, nq - 1);
    const int ai = a0 + (IS_LIN ? qc / DK : qc);
    const int k = active[ai];
    return IS_LIN ? (size_t)k * DK + (qc - (ai - a0) * DK) : (size_t)k;
  };
  auto fetch = [&](int st) {
    const int q0 = st * FB_CG_KC;
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
      const int piece = tid + 256 * u, r = piece >> 6, c = (piece & 63) * 2;
      const size_t prow = row_of(q0 + r);
      const double *src = P + prow * N;
      const double x = src[min(n0 + c, N - 1)], y = src[min(n0 + c + 1, N - 1)];  // clamped, then masked
      const bool rok = q0 + r < nq;
      const double2 v = make_double2((rok && n0 + c < N) ? x : 0.0, (rok && n0 + c + 1 < N) ? y : 0.0);
      rb[u] = v;
    }
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int piece = tid + 256 * u, lr = piece >> 5, lc = (piece & 31) * 2;
      const size_t prow = row_of(q0 + lr);
      const double *src = AT + prow * ldA;
      const double x = src[min(b0 + lc, ldA - 1)], y = src[min(b0 + lc + 1, ldA - 1)];
      const bool rok = q0 + lr < nq;
      ra[u] = make_double2((rok && b0 + lc < ldA) ? x : 0.0, (rok && b0 + lc + 1 < ldA) ? y : 0.0);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
      const int piece = tid + 256 * u, r = piece >> 6, c = (piece & 63) * 2;
      *reinterpret_cast<double2 *>(&sB[buf][r * FB_CG_LDB + c]) = rb[u];
    }
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int piece = tid + 256 * u, lr = piece >> 5, lc = (piece & 31) * 2;
      *reinterpret_cast<double2 *>(&sA[buf][lr * FB_CG_LDA + lc]) = ra[u];
    }
  };
  fb_d4 acc[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[c][t] = fb_d4{0.0, 0.0, 0.0, 0.0};
  if (nst > 0) {
    fetch(0);
    park(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      const int buf = st & 1;
      if (st + 1 < nst) fetch(st + 1);
#pragma unroll
      for (int ks = 0; ks < FB_CG_KC / 4; ++ks) {
        const int kk = 4 * ks + (lane >> 4);
        double af[4], bf[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = sA[buf][kk * FB_CG_LDA + 16 * t + (lane & 15)];
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[c] = sB[buf][kk * FB_CG_LDB + 32 * w + 16 * c + (lane & 15)];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t], bf[c], acc[c][t], 0, 0, 0);
      }
      if (st + 1 < nst) park(buf ^ 1);
      __syncthreads();
    }
  }
  double *o = IS_LIN ? out + (size_t)blockIdx.y * B * N : out;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = n0 + 32 * w + 16 * c + (lane & 15);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = b0 + 16 * t + (lane >> 4) + 4 * i;
        if (row < B && col < N) o[(size_t)row * N + col] = acc[c][t][i];
      }
  }
}

// LDS-DMA form of the same GEMM (default): the register-staged kernel above keeps one stage (8-16 KB per workgroup) in
// flight, which is far too little to cover HBM latency -- 1.9 TB/s on the parameter stream (115 + 60 us; this
// kernel: 80 + 44 us = 2.8 TB/s and 56 % of the f64 MFMA peak on the quadratic term).  Here every 1 KB row
// segment goes straight into LDS with one global_load_lds_dwordx4 per wave, into a ring of FB_CD_S stages filled
// FB_CD_S - 1 stages ahead and awaited with a counted s_waitcnt (hipcc does not see these loads).  Source addresses
// are per lane, so the ragged last column block just clamps its column index (those columns are never stored), and
// padding rows of the last K stage read the always-zero row the engine keeps behind gammaT / XT.  Coefficient rows
// (64 utterances = 512 B) travel two per instruction: LDS row pair p holds K rows p and p + KC/2, which puts the 4
// rows of one MFMA fragment in 4 different pairs, 1152 B apart (bank offset 32 of 64).
#define FB_CD_KC 8
#ifndef FB_CD_S
#define FB_CD_S 3  // 41 KB of LDS: 3 workgroups per CU, all 627 column tiles of the quadratic term resident at once
#endif
#define FB_CD_LDP 144  // doubles per staged coefficient row PAIR (2 x 64 + 16)
__device__ __forceinline__ void fb_iv_glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef double fb_cd_sa_t[FB_CD_S][(FB_CD_KC / 2) * FB_CD_LDP];
typedef double fb_cd_sb_t[FB_CD_S][FB_CD_KC * FB_CG_LDB];
// (the body of one workgroup; bx / by / bz = its tile, K chunk and utterance group: a launch index of its own kernel or
// decoded from the joint launch's, k_iv_contract_both)
template <bool IS_LIN>
__device__ __forceinline__ void fb_contract_dma_body(const FbIvDev &iv, const double *__restrict__ AT, int ldA, size_t zero_row,
                                                     const int *__restrict__ active, const int *__restrict__ n_active, int B,
                                                     int n_kchunks, double *__restrict__ out, const int bx, const int by,
                                                     const int bz, fb_cd_sa_t &sA, fb_cd_sb_t &sB) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = IS_LIN ? iv.R : iv.triR;
  const int DK = IS_LIN ? iv.D : 1;
  const double *P = IS_LIN ? iv.sim : iv.u;
  const int n0 = bx * FB_CG_NT;
  const int b0 = bz * 64;
  const int na = *n_active;
  int a0 = 0, a1 = na;
  if (IS_LIN) {
    const int per = (na + n_kchunks - 1) / n_kchunks;
    a0 = by * per;
    a1 = min(na, a0 + per);
  }
  const int nq = max(0, a1 - a0) * DK;
  const int nst = (nq + FB_CD_KC - 1) / FB_CD_KC;
  const unsigned sA_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)&sA[0][0];
  const unsigned sB_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)&sB[0][0];
  // per-lane column offsets (clamped: the ragged last block re-reads valid columns it never stores)
  const int colB = min(n0 + 2 * lane, N - 2);
  const int colA = min(b0 + 2 * (lane & 31), ldA - 2);
  auto row_of = [&](int q, bool &valid) -> size_t {   // parameter / coefficient row of contraction index q (wave-uniform)
    valid = q < nq;
    const int qc = min(q, max(nq - 1, 0));
    const int ai = a0 + (IS_LIN ? qc / DK : qc);
    const int k = active[ai];
    return IS_LIN ? (size_t)k * DK + (qc - (ai - a0) * DK) : (size_t)k;
  };
  // stage st -> ring slot st % S: this wave brings parameter rows w and w + 4 and coefficient row pair w
  auto issue = [&](int st) {
    const int slot = st % FB_CD_S, q0 = st * FB_CD_KC;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bool ok;
      const size_t prow = row_of(q0 + w + 4 * u, ok);
      fb_iv_glds16(P + prow * N + colB, sB_lds + (unsigned)((slot * FB_CD_KC + w + 4 * u) * FB_CG_LDB * 8));
    }
    bool ok0, ok1;
    const size_t r0 = row_of(q0 + w, ok0), r1 = row_of(q0 + w + FB_CD_KC / 2, ok1);
    const size_t ra = ok0 ? r0 : zero_row, rb = ok1 ? r1 : zero_row;
    const double *src = AT + (lane < 32 ? ra : rb) * ldA + colA;
    fb_iv_glds16(src, sA_lds + (unsigned)((slot * (FB_CD_KC / 2) + w) * FB_CD_LDP * 8));
  };
  fb_d4 acc[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[c][t] = fb_d4{0.0, 0.0, 0.0, 0.0};
  if (nst > 0) {
#pragma unroll
    for (int p = 0; p < FB_CD_S - 1; ++p) issue(p);     // stages beyond nst re-read clamped rows into unused slots
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (FB_CD_S - 2)) : "memory");  // stage 0 has landed (this wave's pieces)
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      const int slot = st % FB_CD_S;
      issue(st + FB_CD_S - 1);  // its slot was read in iteration st - 1
#pragma unroll
      for (int ks = 0; ks < FB_CD_KC / 4; ++ks) {
        const int kk = 4 * ks + (lane >> 4);
        double af[4], bf[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = sA[slot][(kk & 3) * FB_CD_LDP + (kk >> 2) * 64 + 16 * t + (lane & 15)];
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[c] = sB[slot][kk * FB_CG_LDB + 32 * w + 16 * c + (lane & 15)];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t], bf[c], acc[c][t], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (FB_CD_S - 2)) : "memory");  // stage st + 1 has landed
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  double *o = IS_LIN ? out + (size_t)by * B * N : out;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = n0 + 32 * w + 16 * c + (lane & 15);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = b0 + 16 * t + (lane >> 4) + 4 * i;
        if (row < B && col < N) o[(size_t)row * N + col] = acc[c][t][i];
      }
  }
}

template <bool IS_LIN>
__global__ __launch_bounds__(256) void k_iv_contract_dma(FbIvDev iv, const double *__restrict__ AT, int ldA, size_t zero_row,
                                                         const int *__restrict__ active,
                                                         const int *__restrict__ n_active, int B, int n_kchunks,
                                                         double *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) fb_cd_sa_t sA;
  __shared__ __attribute__((aligned(16))) fb_cd_sb_t sB;
  fb_contract_dma_body<IS_LIN>(iv, AT, ldA, zero_row, active, n_active, B, n_kchunks, out, blockIdx.x, blockIdx.y, blockIdx.z, sA, sB);
}
// Both contractions in ONE launch (round 4): the linear one (Sigma^-1 M: few flops, bound by the latency of its short K
// chunks) first in launch order, the quadratic one (U: bound by the float64 matrix pipe) behind it -- the workgroups of
// the second fill the compute units beside those of the first instead of waiting for its launch to drain.
__global__ __launch_bounds__(256) void k_iv_contract_both(FbIvDev iv, const double *__restrict__ XT,
                                                          const double *__restrict__ gammaT, int ldA,
                                                          const int *__restrict__ active, const int *__restrict__ n_active,
                                                          int B, int n_kchunks, double *__restrict__ linp,
                                                          double *__restrict__ quad, int n_lin_x, int n_quad_x) {
  __shared__ __attribute__((aligned(16))) fb_cd_sa_t sA;
  __shared__ __attribute__((aligned(16))) fb_cd_sb_t sB;
  const int n_lin = n_lin_x * n_kchunks * (int)gridDim.y;
  int idx = blockIdx.x;
  if (idx < n_lin_x * n_kchunks) {
    // The column tiles of one K chunk read the SAME coefficient rows (X^T: 13 MB over all chunks at configs[2] size).
    // Workgroups are dealt round-robin over the eight XCDs, each with an L2 of its own: with tile = idx % n_lin_x the four
    // tiles of a chunk sat on four XCDs and each fetched the rows again.  Here a chunk's tiles are the workgroups idx,
    // idx + 8, idx + 16, ... -- one XCD (chunk counts that are not a multiple of 8 keep the plain order).
    int tile = idx % n_lin_x, chunk = idx / n_lin_x;
    if ((n_kchunks & 7) == 0) {
      const int grp = idx / (8 * n_lin_x), in = idx - grp * 8 * n_lin_x;
      chunk = grp * 8 + (in & 7);
      tile = in >> 3;
    }
    fb_contract_dma_body<true>(iv, XT, ldA, (size_t)iv.C * iv.D, active, n_active, B, n_kchunks, linp, tile, chunk,
                               blockIdx.y, sA, sB);
  } else {
    idx -= n_lin_x * n_kchunks;
    fb_contract_dma_body<false>(iv, gammaT, ldA, (size_t)iv.C, active, n_active, B, 1, quad, idx, 0, blockIdx.y, sA, sB);
  }
  (void)n_lin; (void)n_quad_x;
}

void fb_launch_iv_contract(hipStream_t s, const FbIvDev &iv, const double *gammaT, const double *XT, int B,
                           int Bpad, int n_kchunks, int *flags, int *active, int *n_active, double *linp,
                           double *quad, int *fail) {
  hipLaunchKernelGGL(k_iv_active, dim3(1), dim3(1024), 0, s, iv.C, flags, active, n_active, fail);
  const int bgroups = (B + 63) / 64;
  // LDS-DMA form: needs 16-byte aligned row segments; odd R / odd R(R+1)/2 take the register-staged kernel
  const bool dma = (iv.R % 2 == 0) && (iv.triR % 2 == 0) && iv.R >= 2;
  static const bool split = getenv("FB_IV_CONTRACT_SPLIT") != nullptr;  // A/B: the two launches
  if (dma && !split) {
    const int n_lin_x = (iv.R + FB_CG_NT - 1) / FB_CG_NT, n_quad_x = (iv.triR + FB_CG_NT - 1) / FB_CG_NT;
    hipLaunchKernelGGL(k_iv_contract_both, dim3(n_lin_x * n_kchunks + n_quad_x, bgroups), dim3(256), 0, s, iv, XT, gammaT, Bpad,
                       active, n_active, B, n_kchunks, linp, quad, n_lin_x, n_quad_x);
    return;
  }
  if (dma) {
    hipLaunchKernelGGL((k_iv_contract_dma<true>), dim3((iv.R + FB_CG_NT - 1) / FB_CG_NT, n_kchunks, bgroups), dim3(256), 0, s,
                       iv, XT, Bpad, (size_t)iv.C * iv.D, active, n_active, B, n_kchunks, linp);
    hipLaunchKernelGGL((k_iv_contract_dma<false>), dim3((iv.triR + FB_CG_NT - 1) / FB_CG_NT, 1, bgroups), dim3(256), 0, s,
                       iv, gammaT, Bpad, (size_t)iv.C, active, n_active, B, 1, quad);
    return;
  }
  hipLaunchKernelGGL((k_iv_contract_gemm<true, 16>), dim3((iv.R + FB_CG_NT - 1) / FB_CG_NT, n_kchunks, bgroups), dim3(256), 0,
                     s, iv, XT, Bpad, active, n_active, B, n_kchunks, linp);
  hipLaunchKernelGGL((k_iv_contract_gemm<false, 8>), dim3((iv.triR + FB_CG_NT - 1) / FB_CG_NT, 1, bgroups), dim3(256), 0, s,
                     iv, gammaT, Bpad, active, n_active, B, 1, quad);
}

// (K10c, the posterior systems: ivector_solve.hip)

// ------------------------------------------------------ back-end (K11/K12)
// One workgroup of 1024 threads per utterance: fb_iv_backend_body (fb_iv_tail.h) as a kernel of its own -- what runs when
// the solve kernels' tail does not take the back-end along (FB_IV_TAIL=split, A/B and tests).  The two mat-vecs (LDA: L x
// R, PLDA transform: L x L) are split four ways along the contraction index -- thread (g, t) sums a quarter of the
// products of output t, the quarters are added in fixed order -- so that a lane's dependent chain is R/4 loads long
// instead of R, with 20 - 25 of them in flight (the kernel is a latency chain: 77 us with 256 threads and whole dot
// products per thread, 35 with 8 loads in flight, 30 with 20; five shares over 1000 threads measured the same as four).
__global__ __launch_bounds__(1024) void k_iv_backend(FbIvDev iv, const double *__restrict__ ivec,
                                                     double *__restrict__ llr) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  const int R = iv.R, b = blockIdx.x;
  for (int r = threadIdx.x; r < R; r += 1024) smd[r] = (double)(float)ivec[(size_t)b * R + r] - iv.mean_vec[r];
  __syncthreads();
  fb_iv_backend_body<1024>(iv, b, smd, llr, false);
}
void fb_launch_iv_backend(hipStream_t s, const FbIvDev &iv, const double *ivec, int B, double *llr) {
  const size_t shm = sizeof(double) * (size_t)fb_ivt_backend_doubles(iv.R, iv.L);
  hipLaunchKernelGGL(k_iv_backend, dim3(B), dim3(1024), shm, s, iv, ivec, llr);
}
size_t fb_iv_tail_lds_doubles(const FbIvDev &iv) { return (size_t)fb_ivt_backend_doubles(iv.R, iv.L) + FB_LOSS_LDS + FB_SC_LDS; }
bool fb_iv_tail_takes_loss(int B) { return B - 1 <= 128 && B <= FB_LOSS_LDS; }
