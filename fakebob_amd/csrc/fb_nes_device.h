// fb_nes_device.h -- device code shared by the NES kernels (nes_kernels.hip) and the fused GMM finalisation
// (gmm_kernels.hip): numpy-order summation and the loss / loop-control body of FakeBob.loss_fn + attack.
#pragma once
#include "fb_device.h"
#include "fb_kernels.h"

__device__ __forceinline__ double fb_ld_agent_f64(const double *p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), FB_XCH_LD,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}

// numpy pairwise summation over elem(i), i in [lo, lo+n)  (see oracle/fb_oracle.c
// fbo_np_sum; verified bit-for-bit against numpy 2.2.6)
struct D1 {
  double v;
  __device__ __forceinline__ static D1 zero() { return D1{0.0}; }
  __device__ __forceinline__ D1 operator+(const D1 &o) const { return D1{__dadd_rn(v, o.v)}; }
};
struct D4 {
  double v[4];
  __device__ __forceinline__ static D4 zero() { return D4{{0.0, 0.0, 0.0, 0.0}}; }
  __device__ __forceinline__ D4 operator+(const D4 &o) const {
    return D4{{__dadd_rn(v[0], o.v[0]), __dadd_rn(v[1], o.v[1]), __dadd_rn(v[2], o.v[2]), __dadd_rn(v[3], o.v[3])}};
  }
};
template <typename T, typename F>
__device__ __forceinline__ T fb_np_sum_block(F elem, int lo, int n) {  // n <= 128
  if (n < 8) {
    T r = T::zero();
    for (int i = 0; i < n; ++i) r = r + elem(lo + i);
    return r;
  }
  T r0 = elem(lo + 0), r1 = elem(lo + 1), r2 = elem(lo + 2), r3 = elem(lo + 3);
  T r4 = elem(lo + 4), r5 = elem(lo + 5), r6 = elem(lo + 6), r7 = elem(lo + 7);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 = r0 + elem(lo + i + 0); r1 = r1 + elem(lo + i + 1); r2 = r2 + elem(lo + i + 2);
    r3 = r3 + elem(lo + i + 3); r4 = r4 + elem(lo + i + 4); r5 = r5 + elem(lo + i + 5);
    r6 = r6 + elem(lo + i + 6); r7 = r7 + elem(lo + i + 7);
  }
  T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res = res + elem(lo + i);
  return res;
}
// numpy recurses sum(lo,n) = sum(lo,n2) + sum(lo+n2,n-n2), n2 = n/2 - (n/2)%8, above 128
// elements; an explicit post-order stack replaces the recursion (no device call stack).
template <typename T, typename F>
__device__ __forceinline__ T fb_np_sum(F elem, int lo, int n) {
  if (n <= 128) return fb_np_sum_block<T>(elem, lo, n);
  int lo_s[14], n_s[14], st_s[14];
  T left_s[14];
  int sp = 0;
  T ret = T::zero();
  lo_s[0] = lo; n_s[0] = n; st_s[0] = 0;
  while (sp >= 0) {
    const int cn = n_s[sp], cl = lo_s[sp];
    if (cn <= 128) {
      ret = fb_np_sum_block<T>(elem, cl, cn);
      --sp;
    } else {
      int n2 = cn / 2;
      n2 -= n2 % 8;
      if (st_s[sp] == 0) {
        st_s[sp] = 1;
        ++sp; lo_s[sp] = cl; n_s[sp] = n2; st_s[sp] = 0;
      } else if (st_s[sp] == 1) {
        left_s[sp] = ret;
        st_s[sp] = 2;
        ++sp; lo_s[sp] = cl + n2; n_s[sp] = cn - n2; st_s[sp] = 0;
      } else {
        ret = left_s[sp] + ret;
        --sp;
      }
    }
  }
  return ret;
}

#define FB_LOSS_LDS 1024
#define FB_SC_LDS 2048
// SMALL: samples_per_draw <= 128 -- numpy's sum is a single block then and the kernel needs no
// recursion stack (the stack lives in scratch memory, which also slows the dispatch down)
// FUSED: the caller is the last workgroup of k_gmm_finalize_loss; raw[] was written by OTHER workgroups of the same
// launch and is read with agent-scope loads.
#ifdef FB_FIN_STAMP  // instrumented build of k_gmm_finalize_loss_update (tools/profile/fin_instrumented.sh): only gmm_kernels.hip is built with it
static __device__ unsigned long long g_fin_stamps[1024 * 12];
#define FN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_fin_stamps[blockIdx.x * 12 + (k)] = wall_clock64(); } while (0)
#else
#define FN_STAMP(k) do { } while (0)
#endif
template <bool SMALL, bool FUSED>
__device__ __forceinline__ void fb_loss_body(const double *__restrict__ raw, const int *__restrict__ tv,
                                              int B, int M, int task, int znorm_all, int attack_type,
                                              const double *__restrict__ z_mean,
                                              const double *__restrict__ z_std, double threshold,
                                              double adver_thresh, int target, int true_label,
                                              const double *__restrict__ dist_part, int n_dist_part,
                                              double *__restrict__ scores, double *__restrict__ loss,
                                              FbNesDev *__restrict__ out, FbCtlDev *__restrict__ ctl,
                                              double *__restrict__ trace, int it, double *__restrict__ s_lv,
                                              double *__restrict__ s_sc, const int pub_seq = 0,
                                              const int lv_cap = FB_LOSS_LDS, const int sc_cap = FB_SC_LDS) {
  // pub_seq != 0: workgroups of the SAME launch wait for this body's results (k_gmm_finalize_loss_update's update part):
  // the losses, the step size and the stop flag go out as agent-scope (write-through) stores, and when they are complete
  // ctl->pub_seq = pub_seq tells the pollers -- no device-wide fence
  const int S = (task == FB_TASK_CSI || znorm_all) ? M : M - 1;
  __shared__ int s_errw[16];
  // The decisions at the end are one thread's work: everything it needs from global memory is requested HERE and
  // arrives while the block computes the losses; nothing below waits for it before it is used (no barrier up here: the
  // "no voiced frames" flag is reduced over the waves at the end instead of being initialised in LDS first).  Read
  // where they were used, the control block, the window of recent losses, the raw scores (a run-time loop: one L2
  // round trip per model) and the losses were ~20 dependent round trips, 9 of the fused kernel's 18 us.
  // s_lv[FB_LOSS_LDS]: the losses of this iteration (B <= FB_LOSS_LDS; otherwise read back from `loss`); s_sc[FB_SC_LDS]: the
  // scores while the loss is formed from them (B S <= FB_SC_LDS; otherwise in `scores`) -- LDS of the caller (static
  // arrays in the small kernels, a piece of the dynamic allocation in the solve kernels' tail)
  constexpr int FB_LS_LOCAL = 8;
  const bool sc_lds = (size_t)B * S <= (size_t)sc_cap;
  const double dist_first = (int)threadIdx.x < n_dist_part ? dist_part[threadIdx.x] : 0.0;  // in flight with the rest
  FbCtlDev c = {};
  double lsv[FB_LS_LOCAL] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (threadIdx.x == 0 && ctl) c = *ctl;
  int my_err = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (tv && tv[b] <= 0) my_err = b + 1 > my_err ? b + 1 : my_err;
    constexpr int FB_RAW_LOCAL = 8;  // all raw scores of the utterance in one batch of loads
    double rv[FB_RAW_LOCAL];
    if (M <= FB_RAW_LOCAL) {
#pragma unroll
      for (int m = 0; m < FB_RAW_LOCAL; ++m) {
        const double *rp = raw + (size_t)b * M + (m < M ? m : M - 1);
        rv[m] = FUSED ? fb_ld_agent_f64(rp) : *rp;
      }
    }
    auto r = [&](int m) -> double {
      if (M <= FB_RAW_LOCAL) {
        double v = rv[0];
#pragma unroll
        for (int q = 1; q < FB_RAW_LOCAL; ++q) v = m == q ? rv[q] : v;
        return v;
      }
      return FUSED ? fb_ld_agent_f64(raw + (size_t)b * M + m) : raw[(size_t)b * M + m];
    };
    double *sc = sc_lds ? s_sc + (size_t)b * S : scores + (size_t)b * S;
#ifdef FB_FIN_STAMP
    if (rv[0] == 12345.678) FN_STAMP(11); else FN_STAMP(11);  // (raw scores arrived)
#endif
    double l;
    if (M <= FB_RAW_LOCAL) {
      // at most eight scores: they stay in registers (the general path below writes them to LDS and reads every one back
      // through run-time loops -- a dozen dependent LDS round trips in the one workgroup everybody waits for); the same
      // operations on the same values in the same order
      double scr[FB_RAW_LOCAL];
      if (task == FB_TASK_CSI || znorm_all) {
        double zm[FB_RAW_LOCAL], zs[FB_RAW_LOCAL];
#pragma unroll
        for (int m = 0; m < FB_RAW_LOCAL; ++m) { zm[m] = z_mean[m < M ? m : M - 1]; zs[m] = z_std[m < M ? m : M - 1]; }
#pragma unroll
        for (int m = 0; m < FB_RAW_LOCAL; ++m) scr[m] = __ddiv_rn(__dsub_rn(rv[m], zm[m]), zs[m]);
      } else {
#pragma unroll
        for (int m = 0; m < FB_RAW_LOCAL; ++m) scr[m] = __dsub_rn(rv[m + 1 < FB_RAW_LOCAL ? m + 1 : FB_RAW_LOCAL - 1], rv[0]);
      }
#pragma unroll
      for (int m = 0; m < FB_RAW_LOCAL; ++m)
        if (m < S) {
          sc[m] = scr[m];
          if (sc_lds) scores[(size_t)b * S + m] = scr[m];
        }
      auto pick = [&](int i) -> double {
        double v = scr[0];
#pragma unroll
        for (int q = 1; q < FB_RAW_LOCAL; ++q) v = i == q ? scr[q] : v;
        return v;
      };
      auto max_but = [&](int skip) -> double {   // max over m < S, m != skip, in index order
        double om = -INFINITY;
#pragma unroll
        for (int m = 0; m < FB_RAW_LOCAL; ++m)
          if (m < S && m != skip) om = scr[m] > om ? scr[m] : om;
        return om;
      };
      if (task == FB_TASK_SV) {
        l = __dsub_rn(__dadd_rn(threshold, adver_thresh), scr[0]);  // FAKEBOB.py:297
      } else if (task == FB_TASK_OSI && attack_type == FB_UNTARGETED) {
        l = __dsub_rn(__dadd_rn(threshold, adver_thresh), max_but(-1));  // :269
      } else if (task == FB_TASK_OSI) {
        const double om = max_but(target);
        const double mx = om > threshold ? om : threshold;
        l = __dsub_rn(__dadd_rn(mx, adver_thresh), pick(target));  // :262
      } else if (attack_type == FB_TARGETED) {
        l = __dsub_rn(__dadd_rn(max_but(target), adver_thresh), pick(target));  // :281
      } else {
        l = __dsub_rn(__dadd_rn(pick(true_label), adver_thresh), max_but(true_label));  // :291
      }
    } else {
      if (task == FB_TASK_CSI || znorm_all) {
        // gmm_ubm_CSI.py:93; ivector_PLDA_OSI.py:119 / _CSI.py:118 / _SV.py:85
        for (int m = 0; m < M; ++m) sc[m] = __ddiv_rn(__dsub_rn(r(m), z_mean[m]), z_std[m]);
      } else {
        const double r_ubm = r(0);
        for (int m = 0; m < S; ++m) sc[m] = __dsub_rn(r(1 + m), r_ubm);  // gmm_ubm_OSI.py:89, gmm_ubm_SV.py:77
      }
      if (sc_lds) for (int m = 0; m < S; ++m) scores[(size_t)b * S + m] = sc[m];
      if (task == FB_TASK_SV) {
        l = __dsub_rn(__dadd_rn(threshold, adver_thresh), sc[0]);  // FAKEBOB.py:297
      } else if (task == FB_TASK_OSI && attack_type == FB_UNTARGETED) {
        double mx = -INFINITY;
        for (int m = 0; m < S; ++m) mx = sc[m] > mx ? sc[m] : mx;
        l = __dsub_rn(__dadd_rn(threshold, adver_thresh), mx);  // :269
      } else if (task == FB_TASK_OSI) {
        double om = -INFINITY;
        for (int m = 0; m < S; ++m) if (m != target) om = sc[m] > om ? sc[m] : om;
        double mx = om > threshold ? om : threshold;
        l = __dsub_rn(__dadd_rn(mx, adver_thresh), sc[target]);  // :262
      } else if (attack_type == FB_TARGETED) {
        double om = -INFINITY;
        for (int m = 0; m < S; ++m) if (m != target) om = sc[m] > om ? sc[m] : om;
        l = __dsub_rn(__dadd_rn(om, adver_thresh), sc[target]);  // :281
      } else {
        double om = -INFINITY;
        for (int m = 0; m < S; ++m) if (m != true_label) om = sc[m] > om ? sc[m] : om;
        l = __dsub_rn(__dadd_rn(sc[true_label], adver_thresh), om);  // :291
      }
    }
    if (pub_seq) __hip_atomic_store(reinterpret_cast<unsigned long long *>(loss + b), (unsigned long long)__double_as_longlong(l),
                                    FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    else loss[b] = l;
    if (b < lv_cap) s_lv[b] = l;
  }
  // (the window of recent losses hangs off a pointer IN the control block: requested here, behind the raw scores, it
  //  arrives during the barrier and the mean below; requested at the top it put a second round trip in front of them)
  if (threadIdx.x == 0 && ctl && c.plateau_length > 0 && c.plateau_length <= FB_LS_LOCAL) {
#pragma unroll
    for (int i = 0; i < FB_LS_LOCAL; ++i) lsv[i] = c.ls[i < c.plateau_length ? i : c.plateau_length - 1];
  }
  FN_STAMP(6);
  // max |audio - adver| over the perturb kernel's per-workgroup partials (order-independent: every thread takes a
  // strided share instead of thread 0 walking up to N / 256 entries alone)
  __shared__ double s_dmax[16];   // one per wave (the solve kernels' tail calls this body with 512 threads)
  {
    double dm = dist_first > 0.0 ? dist_first : 0.0;
    for (int i = threadIdx.x + blockDim.x; i < n_dist_part; i += blockDim.x) { const double v = dist_part[i]; dm = v > dm ? v : dm; }
    dm = fb_wave_max(dm);
    if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 16) s_dmax[threadIdx.x >> 6] = dm;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(my_err, o, 64); my_err = v > my_err ? v : my_err; }
    if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 16) s_errw[threadIdx.x >> 6] = my_err;
  }
  // (pub_seq: a thread's loss stores must be complete before thread 0 publishes -- every thread waits for its own BEHIND
  //  this barrier, while thread 0 does the serial part below, and a second barrier precedes the publication: the two
  //  store round trips -- the threads' and thread 0's -- overlap each other and the arithmetic instead of adding up)
  __syncthreads();
  FN_STAMP(7);
  // Thread 0: what the pollers of a fused launch wait for -- the stop flag and the step size -- is decided and stored
  // FIRST and published; the bookkeeping (outputs, the window of losses, the trace row, counters) follows the
  // publication: 1.5 us of one lane's stores off the path of the 188 workgroups that wait.
  double al = 0.0, final_loss = 0.0, d = 0.0, lr = c.lr;
  int s_err = 0, n_ls_new = c.n_ls;
  bool stop_now = false, broke = false, window = false;
  if (threadIdx.x == 0) {
    const int spd = B - 1;
    const bool lds_l = B <= lv_cap;
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6) && i < 16; ++i) s_err = s_errw[i] > s_err ? s_errw[i] : s_err;
    al = lds_l ? s_lv[0] : loss[0];
    double lsum;
    if (lds_l) {
      auto el = [&](int i) { return D1{s_lv[1 + i]}; };
      lsum = SMALL ? fb_np_sum_block<D1>(el, 0, spd).v : fb_np_sum<D1>(el, 0, spd).v;
    } else {
      auto el = [&](int i) { return D1{loss[1 + i]}; };
      lsum = SMALL ? fb_np_sum_block<D1>(el, 0, spd).v : fb_np_sum<D1>(el, 0, spd).v;
    }
    // np.mean :243 -- of an empty slice when samples_per_draw < 2: NaN, like NumPy
    final_loss = spd > 0 ? __ddiv_rn(lsum, (double)spd) : __longlong_as_double(0x7ff8000000000000ll);
    FN_STAMP(8);
    for (int i = 0; i < (int)(blockDim.x >> 6) && i < 16; ++i) d = s_dmax[i] > d ? s_dmax[i] : d;
    if (ctl) {
      if (s_err) {
        stop_now = true;
      } else if (al < 0.0 && !c.disable_stop) {  // FAKEBOB.py:181 -- break before the learning-rate step
        stop_now = true;
        broke = true;
      } else {  // :195-200
        const int PL = c.plateau_length;
        if (PL > 0) {
          window = true;
          int n = c.n_ls;
          bool up = false;
          if (PL <= FB_LS_LOCAL) {  // the window of recent losses in registers
            if (n < PL) {
#pragma unroll
              for (int i = 0; i < FB_LS_LOCAL; ++i) if (i == n) lsv[i] = final_loss;
              ++n;
            } else {
#pragma unroll
              for (int i = 1; i < FB_LS_LOCAL; ++i) if (i < PL) lsv[i - 1] = lsv[i];
#pragma unroll
              for (int i = 0; i < FB_LS_LOCAL; ++i) if (i == PL - 1) lsv[i] = final_loss;
            }
            double last = lsv[0];
#pragma unroll
            for (int i = 0; i < FB_LS_LOCAL; ++i) if (i == PL - 1) last = lsv[i];
            up = n == PL && last > lsv[0];
          } else {  // (a long window stays in memory)
            if (n < PL) {
              c.ls[n++] = final_loss;
            } else {
              for (int i = 1; i < PL; ++i) c.ls[i - 1] = c.ls[i];
              c.ls[PL - 1] = final_loss;
            }
            up = n == PL && c.ls[PL - 1] > c.ls[0];
          }
          if (up) {
            if (lr > c.min_lr) {
              const double l2 = __ddiv_rn(lr, c.plateau_drop);
              lr = l2 > c.min_lr ? l2 : c.min_lr;
              if (pub_seq) __hip_atomic_store(reinterpret_cast<unsigned long long *>(&ctl->lr), (unsigned long long)__double_as_longlong(lr),
                                              FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
              else ctl->lr = lr;
            }
            n = 0;
          }
          n_ls_new = n;
        }
      }
      if (stop_now) {
        if (pub_seq) __hip_atomic_store(&ctl->stop, 1, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
        else ctl->stop = 1;
      }
    }
  }
  FN_STAMP(9);
  if (pub_seq) {   // everything the pollers read is complete, then the word they poll
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    FN_STAMP(10);
    __syncthreads();
    if (threadIdx.x == 0 && ctl) __hip_atomic_store(&ctl->pub_seq, pub_seq, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {  // the bookkeeping
    out->adver_loss = al;
    out->final_loss = final_loss;
    const double *sc0 = sc_lds ? s_sc : scores;  // row 0 = the clean adver
    for (int m = 0; m < S && m < 62; ++m) out->score0[m] = sc0[m];
    out->distance = d;
    out->err = s_err;
    if (ctl) {
      if (s_err) {
        ctl->err = s_err;
      } else {
        if (broke) {
          ctl->broke = 1;
          ctl->stop_iter = it;
        } else if (window) {
          const int PL = c.plateau_length;
          if (PL <= FB_LS_LOCAL) {
#pragma unroll
            for (int i = 0; i < FB_LS_LOCAL; ++i) if (i < PL) c.ls[i] = lsv[i];
          }
          ctl->n_ls = n_ls_new;
        }
        double *row = trace ? trace + (size_t)it * (3 + S) : nullptr;
        if (row) {
          row[0] = d; row[1] = al; row[2] = lr;
          for (int m = 0; m < S; ++m) row[3 + m] = sc0[m];
        }
        if (c.ticks) c.ticks[it + 1] = wall_clock64();  // [0] = the attack's start (k_stamp)
        ctl->iters_done = it + 1;
      }
    }
  }
}

// k_grad_update (momentum sign step of iteration `iter`) + k_perturb (the batch of iteration iter + 1) for the 256
// consecutive samples of workgroup `bidx` (nes_kernels.hip: k_update_perturb's description).  WAIT (round 5): the
// workgroup is part of the launch that also finalises the GMM scores and runs the loss body (k_gmm_finalize_loss_update,
// gmm_kernels.hip): what does not depend on the losses -- staging this iteration's normals, DRAWING the next iteration's
// (Philox + Box-Muller: most of the kernel's time) -- is done first, then one thread polls the control block until loss body
// number wait_seq has published (or the attack has stopped), and the losses and the step size are read with agent-scope
// loads.  Same arithmetic in the same order either way: trajectories are bit-identical.
template <bool SMALL, bool WAIT>
__device__ __forceinline__ void fb_update_perturb_body(const double *__restrict__ loss, int64_t N, int half, double sigma,
                                                       float *__restrict__ zbuf, double momentum, double one_minus_m,
                                                       double epsilon, const double *__restrict__ audio,
                                                       double *__restrict__ grad_m, double *__restrict__ adver,
                                                       const FbCtlDev *__restrict__ ctl, uint64_t seed, uint32_t next_iter,
                                                       uint32_t stream, int16_t *__restrict__ q, double *__restrict__ dist_part,
                                                       double qscale, const int bidx, const int wait_seq, double *s_loss) {
  FN_STAMP(0);
  const int spd = 2 * half;
  double *s_a = s_loss + spd;
  float *s_z = reinterpret_cast<float *>(s_a + 256);
  constexpr int MAXI = 5;   // (sample quad, pair) items per thread: 64 half / 512, half <= FB_FUSE_MAX_HALF
  float zn[WAIT ? MAXI : 1][4];
  double lr;
  if constexpr (!WAIT) {
    if (ctl->stop) return;
    lr = ctl->lr;
    for (int i = threadIdx.x; i < spd; i += blockDim.x) s_loss[i] = loss[1 + i];
  }
  const int64_t n = (int64_t)bidx * 256 + threadIdx.x;  // phase 1: threads 0 .. 255
  {  // this iteration's normals of the block's 256 samples: sixteen loads in flight per trip (a plain copy loop waits for
     // every load before it issues the next: 12 dependent round trips at samples_per_draw = 50 -- half of k_update_perturb)
    const int total = half * 256;
    for (int e0 = threadIdx.x; e0 < total; e0 += 16 * (int)blockDim.x) {
      float zv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = min(e0 + u * (int)blockDim.x, total - 1);
        const int64_t ns = (int64_t)bidx * 256 + (e & 255);
        zv[u] = zbuf[(int64_t)(e >> 8) * N + (ns < N ? ns : N - 1)];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = e0 + u * (int)blockDim.x;
        if (e < total) s_z[e] = zv[u];
      }
    }
  }
  const int64_t n4_0 = (int64_t)bidx * 64;  // first sample quad of the block
  // (WAIT: the sample's state is requested before the wait as well -- one memory round trip less behind the release)
  double pre_gm = 0.0, pre_a = 0.0, pre_au = 0.0;
  if (WAIT && threadIdx.x < 256 && n < N) { pre_gm = grad_m[n]; pre_a = adver[n]; pre_au = audio[n]; }
  if constexpr (WAIT) {
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      const int idx = (int)threadIdx.x + u * (int)blockDim.x;
      if (idx < 64 * half) fb_noise4(seed, next_iter, stream, (uint32_t)(n4_0 + (idx & 63)), (uint32_t)(idx >> 6), zn[u]);
    }
    __shared__ int s_stop;
    FN_STAMP(1);
    if (threadIdx.x == 0) {
      int st;
      for (;;) {
        st = __hip_atomic_load(&ctl->stop, FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
        if (st || __hip_atomic_load(&ctl->pub_seq, FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT) >= wait_seq) break;
        __builtin_amdgcn_s_sleep(4);
      }
      // (a stop raised by THIS launch's loss body is published before pub_seq: look again behind it)
      s_stop = st | __hip_atomic_load(&ctl->stop, FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
    }
    FN_STAMP(2);
    __syncthreads();
    if (s_stop) return;
    lr = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(&ctl->lr), FB_XCH_LD,
                                                           __HIP_MEMORY_SCOPE_AGENT));
    for (int i = threadIdx.x; i < spd; i += blockDim.x) s_loss[i] = fb_ld_agent_f64(loss + 1 + i);
  }
  __syncthreads();
  FN_STAMP(3);
  double dmax = 0.0;
  const int smp = threadIdx.x & 255;   // the sample of the block this thread works on in phase 1
  auto el = [&](int i) -> D1 {
    const int j = i < half ? i : i - half;
    double z = (double)s_z[j * 256 + smp];
    if (i >= half) z = -z;
    return D1{__dmul_rn(s_loss[i], z)};
  };
  // The fused launch's 512 threads: NumPy's block sum (fb_np_sum_block, 8 <= n <= 128) keeps eight running sums r0 .. r7
  // over the strided terms and adds them as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); thread t < 256 forms the
  // left half for sample t, thread t + 256 the right half, handed over through s_a -- the same additions, half as many
  // per thread, on the path everybody behind the publication is on.
  const bool split = WAIT && SMALL && spd >= 8 && blockDim.x == 512;
  D1 half_sum = D1::zero();
  if (split) {
    const int64_t ns = (int64_t)bidx * 256 + smp;
    if (ns < N) {
      const int o = threadIdx.x >= 256 ? 4 : 0;
      D1 a0 = el(o), a1 = el(o + 1), a2 = el(o + 2), a3 = el(o + 3);
      for (int i = 8; i < spd - (spd % 8); i += 8) {
        a0 = a0 + el(i + o); a1 = a1 + el(i + o + 1); a2 = a2 + el(i + o + 2); a3 = a3 + el(i + o + 3);
      }
      half_sum = (a0 + a1) + (a2 + a3);
    }
    if (threadIdx.x >= 256) s_a[smp] = half_sum.v;
    __syncthreads();
  }
  if (threadIdx.x >= 256) {
    // phase 2 only
  } else if (n < N) {
    double g = __longlong_as_double(0x7ff8000000000000ll);
    if (spd > 0) {
      double gs;
      if (split) {
        D1 res = half_sum + D1{s_a[threadIdx.x]};
        for (int i = spd - (spd % 8); i < spd; ++i) res = res + el(i);
        gs = res.v;
      } else {
        gs = SMALL ? fb_np_sum_block<D1>(el, 0, spd).v : fb_np_sum<D1>(el, 0, spd).v;
      }
      g = __ddiv_rn(__ddiv_rn(gs, (double)spd), sigma);
    }
    double gm = __dadd_rn(__dmul_rn(momentum, WAIT ? pre_gm : grad_m[n]), __dmul_rn(one_minus_m, g));
    grad_m[n] = gm;
    double sg = gm > 0.0 ? 1.0 : (gm < 0.0 ? -1.0 : gm);  // np.sign (0 -> 0, nan -> nan)
    double a = __dsub_rn(WAIT ? pre_a : adver[n], __dmul_rn(lr, sg));
    const double au = WAIT ? pre_au : audio[n];
    double lo = __dsub_rn(au, epsilon), hi = __dadd_rn(au, epsilon);
    lo = lo < -1.0 ? -1.0 : (lo > 1.0 ? 1.0 : lo);  // np.clip(audio -/+ eps, -1, 1)  (:163-164)
    hi = hi < -1.0 ? -1.0 : (hi > 1.0 ? 1.0 : hi);
    a = a < lo ? lo : a;
    a = a > hi ? hi : a;
    adver[n] = a;
    s_a[threadIdx.x] = a;
    q[n] = fb_quantize(a, qscale);  // column 0 of the next batch: the clean adver
    const double d = fabs(__dsub_rn(au, a));
    dmax = d;
  } else {
    s_a[threadIdx.x] = 0.0;
  }
  {  // distance partial of the NEXT iteration's trace row (max |audio - adver|, as k_perturb reports it)
    __shared__ double red[4];
    double m = fb_wave_max(dmax);
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = m;  // waves 0 .. 3 hold the samples
    __syncthreads();  // also: s_a complete
    if (threadIdx.x == 0) {
      double r = red[0];
      for (int w = 1; w < 4; ++w) r = red[w] > r ? red[w] : r;
      dist_part[bidx] = r;
    }
  }
  FN_STAMP(4);
  // ---- phase 2: the perturbed columns of iteration next_iter for this block's samples
  int u = 0;
  for (int idx = threadIdx.x; idx < 64 * half; idx += blockDim.x, ++u) {
    const int n4l = idx & 63, j = idx >> 6;
    const int64_t n0 = (n4_0 + n4l) * 4;
    if (n0 >= N) continue;
    const int cnt = (N - n0) >= 4 ? 4 : (int)(N - n0);
    float zf[4];
    if constexpr (WAIT) {
      // (u is a compile-time constant after unrolling only for u < MAXI: select instead of indexing)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = zn[0][k];
#pragma unroll
        for (int t = 1; t < MAXI; ++t) v = u == t ? zn[t][k] : v;
        zf[k] = v;
      }
    } else {
      fb_noise4(seed, next_iter, stream, (uint32_t)(n4_0 + n4l), (uint32_t)j, zf);
    }
    float *zp = zbuf + (int64_t)j * N + n0;
    int16_t *qp = q + (int64_t)(1 + j) * N + n0;
    int16_t *qm = q + (int64_t)(1 + half + j) * N + n0;
    int16_t vp[4], vm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double a = s_a[4 * n4l + k], z = (double)zf[k];
      vp[k] = fb_quantize(__dadd_rn(__dmul_rn(sigma, z), a), qscale);   // noise_audios = sigma * noise + audio (:237)
      vm[k] = fb_quantize(__dadd_rn(__dmul_rn(sigma, -z), a), qscale);
    }
    if (cnt == 4 && ((N & 3) == 0)) {
      *reinterpret_cast<float4 *>(zp) = make_float4(zf[0], zf[1], zf[2], zf[3]);
      *reinterpret_cast<short4 *>(qp) = make_short4(vp[0], vp[1], vp[2], vp[3]);
      *reinterpret_cast<short4 *>(qm) = make_short4(vm[0], vm[1], vm[2], vm[3]);
    } else {
      for (int k = 0; k < cnt; ++k) { zp[k] = zf[k]; qp[k] = vp[k]; qm[k] = vm[k]; }
    }
  }
  FN_STAMP(5);
}

