// nes_kernels.hip -- NES-side kernels: Philox noise + perturb + int16 quantise
// (K0/K1), loss (K13/K14), gradient estimate + momentum sign-step (K15/K16).
//
// Follows FAKEBOB.py:223-299 (get_grad / loss_fn) and :193-203 (update) of the
// reference; all float64 arithmetic is written with explicit round-to-nearest
// mul/add so the results are bit-identical to NumPy's (no FMA contraction).
#include "fb_device.h"
#include "fb_kernels.h"
#include "fb_nes_device.h"

// ------------------------------------------------------------------ perturb
// grid.x over n4 blocks (4 samples per thread), grid.y over antithetic pairs.
__global__ __launch_bounds__(256) void k_perturb(const double *__restrict__ adver,
                                                 const double *__restrict__ audio, int64_t N,
                                                 int half, double sigma, uint64_t seed,
                                                 uint32_t iter, uint32_t stream,
                                                 const double *__restrict__ noise_pos,
                                                 int16_t *__restrict__ q,
                                                 double *__restrict__ dist_part,
                                                 float *__restrict__ zbuf, const int *__restrict__ stop,
                                                 double qscale /* 2^(bits_per_sample - 1): gmm_ubm_OSI.py:85 */) {
  if (stop && *stop) return;
  const int64_t n4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  const int64_t n0 = n4 * 4;
  double dmax = 0.0;
  if (n0 < N) {
    double a[4];
    const int cnt = (N - n0) >= 4 ? 4 : (int)(N - n0);
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = (k < cnt) ? adver[n0 + k] : 0.0;
    if (half > 0) {
      double z[4];
      if (noise_pos) {
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (k < cnt) ? noise_pos[(n0 + k) * half + j] : 0.0;
      } else {
        float zf[4];
        fb_noise4(seed, iter, stream, (uint32_t)n4, (uint32_t)j, zf);
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (double)zf[k];
        if (zbuf) {  // keep the float32 normals for the gradient kernel (4.8 MB at spd=50, N=48000)
          float *zp = zbuf + (int64_t)j * N + n0;
          if (cnt == 4 && ((N & 3) == 0)) *reinterpret_cast<float4 *>(zp) = make_float4(zf[0], zf[1], zf[2], zf[3]);
          else for (int k = 0; k < cnt; ++k) zp[k] = zf[k];
        }
      }
      int16_t *qp = q + (int64_t)(1 + j) * N + n0;
      int16_t *qm = q + (int64_t)(1 + half + j) * N + n0;
      int16_t vp[4], vm[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // noise_audios = sigma * noise + audio            (FAKEBOB.py:237)
        double xp = __dadd_rn(__dmul_rn(sigma, z[k]), a[k]);
        double xm = __dadd_rn(__dmul_rn(sigma, -z[k]), a[k]);
        vp[k] = fb_quantize(xp, qscale);
        vm[k] = fb_quantize(xm, qscale);
      }
      if (cnt == 4 && ((N & 3) == 0)) {
        *reinterpret_cast<short4 *>(qp) = make_short4(vp[0], vp[1], vp[2], vp[3]);
        *reinterpret_cast<short4 *>(qm) = make_short4(vm[0], vm[1], vm[2], vm[3]);
      } else {
        for (int k = 0; k < cnt; ++k) { qp[k] = vp[k]; qm[k] = vm[k]; }
      }
    }
    if (j == 0) {
      for (int k = 0; k < cnt; ++k) {
        q[n0 + k] = fb_quantize(a[k], qscale);  // column 0: the clean adver
        if (audio) { double d = fabs(__dsub_rn(audio[n0 + k], a[k])); dmax = d > dmax ? d : dmax; }
      }
    }
  }
  if (blockIdx.y == 0 && dist_part) {
    __shared__ double red[4];
    double m = fb_wave_max(dmax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = red[0];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = red[w] > r ? red[w] : r;
      dist_part[blockIdx.x] = r;
    }
  }
}

void fb_launch_perturb(hipStream_t s, const double *adver, const double *audio, int64_t N, int half,
                       double sigma, uint64_t seed, uint32_t iter, uint32_t stream,
                       const double *noise_pos, int16_t *q, double *dist_part, int *n_dist_part, float *zbuf,
                       const int *stop, int bits) {
  int64_t n4 = (N + 3) / 4;
  dim3 grid((unsigned)((n4 + 255) / 256), (unsigned)(half > 0 ? half : 1));
  if (n_dist_part) *n_dist_part = (int)grid.x;
  hipLaunchKernelGGL(k_perturb, grid, dim3(256), 0, s, adver, audio, N, half, sigma, seed, iter, stream,
                     noise_pos, q, dist_part, zbuf, stop, ldexp(1.0, bits - 1));
}

// Foreign-model path (the reference's plugin API, README.md:136: any `model` with score / make_decisions): the
// perturbed batch leaves the device as float64 -- the model does its own int16 cast -- utterance-major:
// x[0] = adver, x[1+j] = adver + sigma z_j, x[1+half+j] = adver - sigma z_j, exactly FAKEBOB.py:234-237.
__global__ __launch_bounds__(256) void k_perturb_f64(const double *__restrict__ adver,
                                                     const double *__restrict__ audio, int64_t N, int half,
                                                     double sigma, uint64_t seed, uint32_t iter, uint32_t stream,
                                                     const double *__restrict__ noise_pos, double *__restrict__ x,
                                                     double *__restrict__ dist_part, float *__restrict__ zbuf) {
  const int64_t n4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  const int64_t n0 = n4 * 4;
  double dmax = 0.0;
  if (n0 < N) {
    const int cnt = (N - n0) >= 4 ? 4 : (int)(N - n0);
    double a[4];
    for (int k = 0; k < 4; ++k) a[k] = (k < cnt) ? adver[n0 + k] : 0.0;
    if (half > 0) {
      double z[4];
      if (noise_pos) {
        for (int k = 0; k < 4; ++k) z[k] = (k < cnt) ? noise_pos[(n0 + k) * half + j] : 0.0;
      } else {
        float zf[4];
        fb_noise4(seed, iter, stream, (uint32_t)n4, (uint32_t)j, zf);
        for (int k = 0; k < 4; ++k) z[k] = (double)zf[k];
        if (zbuf) for (int k = 0; k < cnt; ++k) zbuf[(int64_t)j * N + n0 + k] = zf[k];
      }
      for (int k = 0; k < cnt; ++k) {
        x[(int64_t)(1 + j) * N + n0 + k] = __dadd_rn(__dmul_rn(sigma, z[k]), a[k]);
        x[(int64_t)(1 + half + j) * N + n0 + k] = __dadd_rn(__dmul_rn(sigma, -z[k]), a[k]);
      }
    }
    if (j == 0) {
      for (int k = 0; k < cnt; ++k) {
        x[n0 + k] = a[k];
        if (audio) { double d = fabs(__dsub_rn(audio[n0 + k], a[k])); dmax = d > dmax ? d : dmax; }
      }
    }
  }
  if (blockIdx.y == 0 && dist_part) {
    __shared__ double red[4];
    double m = fb_wave_max(dmax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = red[0];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = red[w] > r ? red[w] : r;
      dist_part[blockIdx.x] = r;
    }
  }
}
void fb_launch_perturb_f64(hipStream_t s, const double *adver, const double *audio, int64_t N, int half,
                           double sigma, uint64_t seed, uint32_t iter, uint32_t stream, const double *noise_pos,
                           double *x, double *dist_part, int *n_dist_part, float *zbuf) {
  int64_t n4 = (N + 3) / 4;
  dim3 grid((unsigned)((n4 + 255) / 256), (unsigned)(half > 0 ? half : 1));
  if (n_dist_part) *n_dist_part = (int)grid.x;
  hipLaunchKernelGGL(k_perturb_f64, grid, dim3(256), 0, s, adver, audio, N, half, sigma, seed, iter, stream,
                     noise_pos, x, dist_part, zbuf);
}

__global__ __launch_bounds__(256) void k_quantize(const double *__restrict__ x, int64_t n, double scale,
                                                  int16_t *__restrict__ q) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) q[i] = fb_quantize(x[i], scale);
}
__global__ void k_stamp(unsigned long long *t) { *t = wall_clock64(); }
void fb_launch_stamp(hipStream_t s, unsigned long long *t) { hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, s, t); }
void fb_launch_quantize(hipStream_t s, const double *x, int64_t n, int bits, int16_t *q) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_quantize, dim3(blocks), dim3(256), 0, s, x, n, ldexp(1.0, bits - 1), q);
}

__global__ __launch_bounds__(256) void k_noise(uint64_t seed, uint32_t iter, uint32_t stream, int64_t N,
                                               int half, float *__restrict__ z) {
  const int64_t n4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (n4 * 4 >= N) return;
  float v[4];
  fb_noise4(seed, iter, stream, (uint32_t)n4, (uint32_t)j, v);
  for (int k = 0; k < 4; ++k)
    if (n4 * 4 + k < N) z[(int64_t)j * N + n4 * 4 + k] = v[k];
}
void fb_launch_noise(hipStream_t s, uint64_t seed, uint32_t iter, uint32_t stream, int64_t N, int half,
                     float *z) {
  int64_t n4 = (N + 3) / 4;
  dim3 grid((unsigned)((n4 + 255) / 256), (unsigned)half);
  hipLaunchKernelGGL(k_noise, grid, dim3(256), 0, s, seed, iter, stream, N, half, z);
}

// --------------------------------------------------------------------- loss
// (numpy-order sums and the loss body: fb_nes_device.h, shared with k_gmm_finalize_loss)
// SMALL: samples_per_draw <= 128 -- numpy's sum is a single block then and the kernel needs no
// recursion stack (the stack lives in scratch memory, which also slows the dispatch down)
template <bool SMALL>
__global__ __launch_bounds__(256) void k_loss(const double *__restrict__ raw, const int *__restrict__ tv,
                                              int B, int M, int task, int znorm_all, int attack_type,
                                              const double *__restrict__ z_mean,
                                              const double *__restrict__ z_std, double threshold,
                                              double adver_thresh, int target, int true_label,
                                              const double *__restrict__ dist_part, int n_dist_part,
                                              double *__restrict__ scores, double *__restrict__ loss,
                                              FbNesDev *__restrict__ out, FbCtlDev *__restrict__ ctl,
                                              double *__restrict__ trace, int it) {
  if (ctl && ctl->stop) return;  // queued behind the stopping iteration
  __shared__ double s_lv[FB_LOSS_LDS], s_sc[FB_SC_LDS];
  fb_loss_body<SMALL, false>(raw, tv, B, M, task, znorm_all, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                             true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it, s_lv, s_sc);
}

void fb_launch_loss(hipStream_t s, const double *raw, const int *tv, int B, int M, int task, int znorm_all,
                    int attack_type, const double *z_mean, const double *z_std, double threshold,
                    double adver_thresh, int target, int true_label, const double *dist_part,
                    int n_dist_part, double *scores, double *loss, FbNesDev *out, FbCtlDev *ctl, double *trace,
                    int it) {
  if (B - 1 <= 128)
    hipLaunchKernelGGL(k_loss<true>, dim3(1), dim3(256), 0, s, raw, tv, B, M, task, znorm_all, attack_type, z_mean,
                       z_std, threshold, adver_thresh, target, true_label, dist_part, n_dist_part, scores, loss, out,
                       ctl, trace, it);
  else
    hipLaunchKernelGGL(k_loss<false>, dim3(1), dim3(256), 0, s, raw, tv, B, M, task, znorm_all, attack_type, z_mean,
                       z_std, threshold, adver_thresh, target, true_label, dist_part, n_dist_part, scores, loss, out,
                       ctl, trace, it);
}

// ------------------------------------------------------------- grad + update
// estimate_grad = np.mean(loss.flatten() * noise, axis=1) / sigma   (FAKEBOB.py:244)
// then grad = m*pre + (1-m)*grad (:193), adver -= lr*sign(grad), clip (:202-203).
// The normals come back from the buffer k_perturb wrote (zbuf, float32 [half][N]) or from the
// caller's float64 tensor (noise_pos [N][half]); products are summed in numpy's pairwise order.
#define FB_GRAD_LDS_PAIRS 150  // normals of a 256-sample block staged in LDS up to samples_per_draw = 300 (past 64 KB of LDS
                               // from 62 pairs on: opt-in).  Unstaged, every term of the gradient sum is a dependent global load:
                               // samples_per_draw = 200 took 65 us
template <bool SMALL>
__global__ __launch_bounds__(256) void k_grad_update(const double *__restrict__ loss, int64_t N, int half,
                                                     double sigma, const float *__restrict__ zbuf,
                                                     const double *__restrict__ noise_pos,
                                                     double *__restrict__ grad_out, int do_update,
                                                     double momentum, double one_minus_m, double lr,
                                                     double epsilon, const double *__restrict__ audio,
                                                     double *__restrict__ grad_m, double *__restrict__ adver,
                                                     const FbCtlDev *__restrict__ ctl) {
  if (ctl) {  // device-controlled attack: the loss kernel decided whether to go on and with which step
    if (ctl->stop) return;
    lr = ctl->lr;
  }
  extern __shared__ double s_loss[];  // loss[1..spd], then (zbuf path) the block's normals [half][256]
  const int spd = 2 * half;
  float *s_z = reinterpret_cast<float *>(s_loss + spd);
  for (int i = threadIdx.x; i < spd; i += blockDim.x) s_loss[i] = loss[1 + i];
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool lds_z = !noise_pos && half <= FB_GRAD_LDS_PAIRS;
  if (lds_z) {  // every load of the thread is in flight before the first use
    const int64_t nc = n < N ? n : N - 1;
    for (int j0 = 0; j0 < half; j0 += 16) {  // sixteen loads in flight per trip (one per trip: `half` dependent round trips)
      float zv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) zv[u] = zbuf[(int64_t)min(j0 + u, half - 1) * N + nc];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (j0 + u < half) s_z[(j0 + u) * 256 + threadIdx.x] = zv[u];
    }
  }
  __syncthreads();
  if (n >= N) return;
  auto el = [&](int i) -> D1 {
    const int j = i < half ? i : i - half;
    double z = noise_pos ? noise_pos[n * half + j]
                         : (lds_z ? (double)s_z[j * 256 + threadIdx.x] : (double)zbuf[(int64_t)j * N + n]);
    if (i >= half) z = -z;
    return D1{__dmul_rn(s_loss[i], z)};
  };
  double g = __longlong_as_double(0x7ff8000000000000ll);  // samples_per_draw < 2: np.mean over an empty axis (:244)
  if (spd > 0) {
    const double gs = SMALL ? fb_np_sum_block<D1>(el, 0, spd).v : fb_np_sum<D1>(el, 0, spd).v;
    g = __ddiv_rn(__ddiv_rn(gs, (double)spd), sigma);
  }
  if (grad_out) grad_out[n] = g;
  if (do_update) {
    double gm = __dadd_rn(__dmul_rn(momentum, grad_m[n]), __dmul_rn(one_minus_m, g));
    grad_m[n] = gm;
    double sg = gm > 0.0 ? 1.0 : (gm < 0.0 ? -1.0 : gm);  // np.sign (0 -> 0, nan -> nan)
    double a = __dsub_rn(adver[n], __dmul_rn(lr, sg));
    double au = audio[n];
    double lo = __dsub_rn(au, epsilon), hi = __dadd_rn(au, epsilon);
    lo = lo < -1.0 ? -1.0 : (lo > 1.0 ? 1.0 : lo);  // np.clip(audio -/+ eps, -1, 1)  (:163-164)
    hi = hi < -1.0 ? -1.0 : (hi > 1.0 ? 1.0 : hi);
    a = a < lo ? lo : a;
    a = a > hi ? hi : a;
    adver[n] = a;
  }
}

// k_grad_update (momentum sign step of iteration `iter`) + k_perturb (the batch of iteration iter + 1) in one launch:
// a workgroup owns 256 consecutive samples.  Phase 1 is k_grad_update for them (normals of iteration `iter` from
// zbuf, products summed in NumPy's pairwise order); the updated samples stay in LDS.  Phase 2 is k_perturb for the
// same samples: the block's 64 x half (sample quad, antithetic pair) items are dealt over its threads, each draws its
// Philox normals for iteration iter + 1, overwrites its zbuf entries (read in phase 1 by this block only) and writes
// the two int16 columns; thread = sample writes the clean column and the distance partial.  Same arithmetic as the
// two kernels, so trajectories are bit-identical.  Device-controlled attacks only (lr / stop from the control block).
// Workgroup = 256 samples (the unit of the distance partials and of the LDS-staged normals) worked on by FB_UP_THREADS
// threads: the update (phase 1) is one thread per sample, the next iteration's columns (phase 2: 64 sample quads x
// spd/2 pairs of Philox + Box-Muller items) are dealt over all of them -- with 256 threads the 188 workgroups of a 3 s
// utterance put fewer waves on the chip than it has SIMDs.  Measured (same box, 1 attack / 3 in flight): 256 threads
// 19.1 us, 4 930 / 7 900 it/s; 512: 5 040 / 7 930; 1024: 13.1 us, 5 110 / 7 300 (the wide workgroups get in the way of
// the other attacks' kernels).
#define FB_UP_THREADS 512
template <bool SMALL>
__global__ __launch_bounds__(FB_UP_THREADS) void k_update_perturb(const double *__restrict__ loss, int64_t N, int half,
                                                        double sigma, float *__restrict__ zbuf, double momentum,
                                                        double one_minus_m, double epsilon,
                                                        const double *__restrict__ audio, double *__restrict__ grad_m,
                                                        double *__restrict__ adver, const FbCtlDev *__restrict__ ctl,
                                                        uint64_t seed, uint32_t next_iter, uint32_t stream,
                                                        int16_t *__restrict__ q, double *__restrict__ dist_part,
                                                        double qscale) {
  extern __shared__ double s_loss[];  // loss[1..spd], the block's updated samples [256], the block's normals [half][256]
  fb_update_perturb_body<SMALL, false>(loss, N, half, sigma, zbuf, momentum, one_minus_m, epsilon, audio, grad_m, adver, ctl, seed,
                                       next_iter, stream, q, dist_part, qscale, (int)blockIdx.x, 0, s_loss);
}
// returns the number of distance partials the launch writes (one per workgroup)
int fb_launch_update_perturb(hipStream_t s, const double *loss, int64_t N, int half, double sigma, float *zbuf,
                             double momentum, double one_minus_m, double epsilon, const double *audio, double *grad_m,
                             double *adver, const FbCtlDev *ctl, uint64_t seed, uint32_t next_iter, uint32_t stream,
                             int16_t *q, double *dist_part, int bits) {
  const int blocks = (int)((N + 255) / 256);
  const double qscale = ldexp(1.0, bits - 1);
  const size_t shm = sizeof(double) * (size_t)(2 * half + 256) + sizeof(float) * 256 * (size_t)(half > 0 ? half : 1);
  const int nthr = FB_UP_THREADS;
  if (2 * half <= 128)
    hipLaunchKernelGGL(k_update_perturb<true>, dim3(blocks), dim3(nthr), shm, s, loss, N, half, sigma, zbuf, momentum,
                       one_minus_m, epsilon, audio, grad_m, adver, ctl, seed, next_iter, stream, q, dist_part, qscale);
  else
    hipLaunchKernelGGL(k_update_perturb<false>, dim3(blocks), dim3(FB_UP_THREADS), shm, s, loss, N, half, sigma, zbuf, momentum,
                       one_minus_m, epsilon, audio, grad_m, adver, ctl, seed, next_iter, stream, q, dist_part, qscale);
  return blocks;
}

void fb_launch_grad_update(hipStream_t s, const double *loss, int64_t N, int half, double sigma,
                           const float *zbuf, const double *noise_pos, double *grad_out, int do_update,
                           double momentum, double one_minus_m, double lr, double epsilon,
                           const double *audio, double *grad_m, double *adver, const FbCtlDev *ctl) {
  int blocks = (int)((N + 255) / 256);
  size_t shm = sizeof(double) * (size_t)(2 * half > 0 ? 2 * half : 1);
  if (!noise_pos && half <= FB_GRAD_LDS_PAIRS) shm += sizeof(float) * 256 * (size_t)half;
  if (shm > 64 * 1024) {
    static std::atomic<unsigned long long> optin{0};
    unsigned long long bit = 0;
    if (fb_device_needs_optin(optin, &bit)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grad_update<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_grad_update<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      optin.fetch_or(bit, std::memory_order_release);
    }
  }
  if (2 * half <= 128)
    hipLaunchKernelGGL(k_grad_update<true>, dim3(blocks), dim3(256), shm, s, loss, N, half, sigma, zbuf, noise_pos,
                       grad_out, do_update, momentum, one_minus_m, lr, epsilon, audio, grad_m, adver, ctl);
  else
    hipLaunchKernelGGL(k_grad_update<false>, dim3(blocks), dim3(256), shm, s, loss, N, half, sigma, zbuf, noise_pos,
                       grad_out, do_update, momentum, one_minus_m, lr, epsilon, audio, grad_m, adver, ctl);
}
