// fb_iv_tail.h -- device code shared by k_iv_backend (ivector_kernels.hip) and the tails of the posterior-solve kernels
// (ivector_solve.hip): the back-end of one utterance and the last-arriver loss body.
//
// Back-end (ivector_PLDA_kaldiHelper.py:262-271): x - mean.vec, LDA (transform-vec), length normalisation,
// Plda::TransformIvector (normalize_length, simple_length_norm = false, n = 1), log-likelihood ratio against every
// enrolled speaker.  The two mat-vecs (LDA: L x R, PLDA transform: L x L) are split FOUR ways along the contraction
// index -- share g of output t is one thread's chain, the four shares are added in fixed order -- whatever the
// workgroup's size: NT = 1024 threads take one share each (k_iv_backend), NT = 512 two each as independent chains (the
// solve kernels' tail).  Same operations in the same order: the results do not depend on who runs the body.
#pragma once
#include "fb_device.h"
#include "fb_kernels.h"
#include "fb_nes_device.h"

__device__ __forceinline__ double fb_ivt_block_sum(double v, double *red) {
  v = fb_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += red[w];
  return r;
}
// doubles of scratch: x[R] z[L] y[L] part[4][L] red[16]
__host__ __device__ __forceinline__ int fb_ivt_backend_doubles(int R, int L) { return R + 6 * L + 16; }

// x[0 .. R) holds ivec - mean.vec on entry (LDS, first R doubles of `scr`); writes llr[b][0 .. S) (agent-scope stores when
// `agent`: another workgroup of the same launch reads them).  All NT threads of the workgroup call.
template <int NT>
__device__ __forceinline__ void fb_iv_backend_body(const FbIvDev &iv, int b, double *__restrict__ scr, double *__restrict__ llr,
                                                   const bool agent) {
  static_assert(NT == 1024 || NT == 512 || NT == 256, "256 threads per share group");
  constexpr int NG = NT / 256, SH = 4 / NG;   // share groups of 256 threads, shares per group
  const int R = iv.R, L = iv.L, S = iv.S, tid = threadIdx.x;
  const int g = tid >> 8, t = tid & 255;
  double *x = scr, *z = x + R, *y = z + L, *part = y + L, *red = part + 4 * L;
  {
    for (int l = t; l < L; l += 256) {
      double acc[SH];
      int r0[SH], r1[SH];
#pragma unroll
      for (int u = 0; u < SH; ++u) {
        const int sh = g * SH + u;
        r0[u] = (int)((long long)R * sh / 4);
        r1[u] = (int)((long long)R * (sh + 1) / 4);
        acc[u] = (sh == 0 && iv.lda_cols == R + 1) ? iv.ldaT[(size_t)R * L + l] : 0.0;
      }
      if constexpr (SH == 1) {
#pragma unroll 20
        for (int r = r0[0]; r < r1[0]; ++r) acc[0] = fma(iv.ldaT[(size_t)r * L + l], x[r], acc[0]);
      } else {
        // the shares of a thread advance together (independent chains: twice the loads in flight); a share's own
        // order is unchanged.  Shares differ in length by at most one row.
        int n = r1[0] - r0[0];
#pragma unroll
        for (int u = 1; u < SH; ++u) n = min(n, r1[u] - r0[u]);
        // (no guards inside: a conditional load is not batched with its neighbours, and the chain of 200 dependent L2
        //  round trips that resulted cost 100 us -- measured)
#pragma unroll 10
        for (int i = 0; i < n; ++i) {
#pragma unroll
          for (int u = 0; u < SH; ++u) acc[u] = fma(iv.ldaT[(size_t)(r0[u] + i) * L + l], x[r0[u] + i], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < SH; ++u)
          for (int r = r0[u] + n; r < r1[u]; ++r) acc[u] = fma(iv.ldaT[(size_t)r * L + l], x[r], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < SH; ++u) part[(g * SH + u) * L + l] = acc[u];
    }
  }
  __syncthreads();
  double nrm = 0.0;
  if (g == 0)
    for (int l = t; l < L; l += 256) {
      const double acc = ((part[l] + part[L + l]) + part[2 * L + l]) + part[3 * L + l];
      z[l] = acc;
      nrm = fma(acc, acc, nrm);
    }
  nrm = sqrt(fb_ivt_block_sum(nrm, red));
  const double ratio = nrm / sqrt((double)L);
  __syncthreads();
  for (int l = tid; l < L; l += NT) z[l] = (ratio != 0.0 ? z[l] / ratio : z[l]) - iv.plda_mean[l];
  __syncthreads();
  {
    for (int l = t; l < L; l += 256) {
      double acc[SH];
      int m0[SH], m1[SH];
#pragma unroll
      for (int u = 0; u < SH; ++u) {
        const int sh = g * SH + u;
        m0[u] = (int)((long long)L * sh / 4);
        m1[u] = (int)((long long)L * (sh + 1) / 4);
        acc[u] = 0.0;
      }
      if constexpr (SH == 1) {
#pragma unroll 25
        for (int m = m0[0]; m < m1[0]; ++m) acc[0] = fma(iv.pldaT[(size_t)m * L + l], z[m], acc[0]);
      } else {
        int n = m1[0] - m0[0];
#pragma unroll
        for (int u = 1; u < SH; ++u) n = min(n, m1[u] - m0[u]);
#pragma unroll 12
        for (int i = 0; i < n; ++i) {
#pragma unroll
          for (int u = 0; u < SH; ++u) acc[u] = fma(iv.pldaT[(size_t)(m0[u] + i) * L + l], z[m0[u] + i], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < SH; ++u)
          for (int m = m0[u] + n; m < m1[u]; ++m) acc[u] = fma(iv.pldaT[(size_t)m * L + l], z[m], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < SH; ++u) part[(g * SH + u) * L + l] = acc[u];
    }
  }
  __syncthreads();
  double dot = 0.0;
  if (g == 0)
    for (int l = t; l < L; l += 256) {
      const double acc = ((part[l] + part[L + l]) + part[2 * L + l]) + part[3 * L + l];
      y[l] = acc;
      dot += acc * acc / (iv.plda_psi[l] + 1.0);
    }
  dot = fb_ivt_block_sum(dot, red);
  const double nf = sqrt((double)L / dot);
  __syncthreads();
  for (int l = tid; l < L; l += NT) y[l] *= nf;
  __syncthreads();
  const double LOG2PI = 1.8378770664093454835606594728112;
  for (int s = 0; s < S; ++s) {
    const double *tr = iv.train + (size_t)s * L;
    double given = 0.0, without = 0.0;
    // (one l per thread while L <= NT: the same partial sums in the same waves as k_iv_backend's 1024 threads, whose
    //  upper waves then add zeros.  The host only lets a 512-thread tail run this for L <= 512 -- fb_engine.hip,
    //  run_scoring --, so "the same bits whichever kernel runs it" holds wherever both can run)
    for (int l = tid; l < L; l += NT) {
      const double psi = iv.plda_psi[l];
      const double mean = psi / (psi + 1.0) * tr[l];
      const double var = 1.0 + psi / (psi + 1.0);
      const double d = y[l] - mean;
      given += log(var) + d * d / var;
      without += log(psi + 1.0) + y[l] * y[l] / (psi + 1.0);
    }
    given = fb_ivt_block_sum(given, red);
    without = fb_ivt_block_sum(without, red);
    if (tid == 0) {
      double sc_ = -0.5 * (given + LOG2PI * L) - (-0.5 * (without + LOG2PI * L));
      if (iv.text_scores) sc_ = fb_round6(sc_);  // ivector-plda-scoring writes text
      if (agent)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(llr + (size_t)b * S + s), (unsigned long long)__double_as_longlong(sc_),
                           FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
      else
        llr[(size_t)b * S + s] = sc_;
    }
  }
}

// The tail of a solve kernel: `sol[0 .. R)` (LDS) = the solution of utterance b's system (the prior offset still in
// element 0: this also writes the i-vector row).  scr: fb_iv_tail_lds_doubles() doubles of LDS that do not overlap sol.
template <int NT>
__device__ __forceinline__ void fb_iv_tail_run(const FbIvDev &iv, const FbIvTail &tl, int b, int B, const double *__restrict__ sol,
                                               double *__restrict__ ivec, double *__restrict__ scr) {
  const int R = iv.R, tid = threadIdx.x;
  for (int r = tid; r < R; r += NT) {
    const double v = sol[r] - (r == 0 ? iv.prior_offset : 0.0);
    ivec[(size_t)b * R + r] = v;
    if (tl.backend) scr[r] = (double)(float)v - iv.mean_vec[r];   // (the i-vector passes through Kaldi's float32 text / ark form)
  }
  if (!tl.backend) return;
  __syncthreads();
  fb_iv_backend_body<NT>(iv, b, scr, tl.llr, tl.loss != 0);
  if (!tl.loss) return;
  __shared__ int s_ivt_last;
  if (tid == 0) {
    // thread 0 stored this utterance's scores with agent-scope (write-through) atomics; they are complete before its own
    // arrival is counted, and the last arriver reads every row with agent-scope loads (k_gmm_finalize_loss's pattern: no
    // device-wide fence)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    s_ivt_last = (atomicAdd(tl.counter, 1) == B - 1);
  }
  __syncthreads();
  if (!s_ivt_last) return;
  if (tid == 0) __hip_atomic_store(tl.counter, 0, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
  if (tl.ctl && tl.ctl->stop) return;   // queued behind the stopping iteration (k_loss's first line)
  double *s_lv = scr + fb_ivt_backend_doubles(R, iv.L), *s_sc = s_lv + FB_LOSS_LDS;
  fb_loss_body<true, true>(tl.llr, tl.tv, B, iv.S, tl.task, 1, tl.attack_type, tl.z_mean, tl.z_std, tl.threshold, tl.adver_thresh,
                           tl.target, tl.true_label, tl.dist_part, tl.n_dist_part, tl.scores, tl.loss_out, tl.out, tl.ctl, tl.trace,
                           tl.it, s_lv, s_sc);
}
