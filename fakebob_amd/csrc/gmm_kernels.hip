// gmm_kernels.hip -- K7: diagonal-GMM frame log-likelihoods for all models of a
// speaker-recognition system, fused with the per-frame logsumexp, on the gfx950
// matrix cores: f32 arithmetic carried by the f16 / bf16 MFMA pipe (k_gmm_fx2w / k_gmm_fx2: two-term f16 split;
// k_gmm_bx3: exact three-term bf16 split for models whose parameters do not fit f16's range).
//
// Replaces `gmm-global-get-frame-likes --average=true MODEL feats` run once per
// model by the reference (gmm_ubm_kaldiHelper.py:202-221) ([EXT] SURVEY.md A.7):
//     ll_k(x) = gconst_k + (mu/var)_k . x - 1/2 (1/var)_k . x^2 ,  ll = logsumexp_k
//
// Mapping (MI355X-first, not a GEMM library call):
//   * D = P x F : "A" operand = parameters of 32 components, "B" operand = 32
//     frames.  With the C/D layout of the 32x32 MFMA every lane then owns ONE
//     frame (col = lane&31) and 16 components of it, so the online logsumexp is
//     lane-local: no cross-lane traffic in the hot loop.
//   * A wave keeps its frames (x and x^2 as split fragments) in registers for the
//     whole kernel; parameter tiles stream HBM/L2 -> LDS and are shared by the 4
//     waves of the workgroup.
//   * Models with bitwise-identical inverse variances (mean-only MAP adaptation,
//     build_spk_models.py:170) share the quadratic term: acc_q = -1/2 iv . x^2 is
//     computed once per tile ("Q item") and every model continues the fma chain
//     from it with its own mu/var ("L item") -- (G + M)*KH MFMAs per tile instead
//     of 2*M*KH.
//   * The component range is split into chunks (grid.y) so the launch has >> 256
//     workgroups; chunk partials (max, sumexp) are merged by k_gmm_finalize, which
//     also performs the voiced-frame average in float64 in a fixed order
//     (deterministic, no atomics).
#include <float.h>

#include "fb_device.h"
#include <cstdlib>
#include "fb_kernels.h"
#include "fb_nes_device.h"
#include "gmm_split.h"

// DUMP = true (template parameter of the kernels below): single-model variant that stores every component
// log-likelihood ll[row][comp] (leading dimension n_tiles*32) instead of reducing them -- the gmm-gselect stage of the
// i-vector path (ivector_kernels.hip) takes the top-n per frame from it, the enrolment statistics use it too.

// ------------------------------------------------------------------------------------------------
// k_gmm_bx3: the computation on the bf16 matrix pipe (16x the f32 MFMA rate) WITHOUT giving up
// f32 accuracy.  Every f32 operand is split exactly into three bf16 terms (v = v1 + v2 + v3, 8
// significant bits each, round-to-nearest residuals -- the three terms carry all 24 bits exactly), and
// a product a*b is accumulated in f32 from the six partial products of order <= 2^-16:
//     a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1        (dropped: a2b3, a3b2, a3b3 <= 2^-24 |ab|)
// Each bf16 x bf16 product is exact in the f32 accumulator, so the result differs from an f32 fma
// chain by the order of f32 additions plus a term below one f32 ulp per product -- the same class of
// difference as between two f32 BLAS implementations (measured against the float64 oracle: §5 of
// DESIGN.md).  K = D is padded to 16*NK; three padding positions carry gconst (its three bf16 terms
// in A's first image against 1.0 in the frame operand), so the epilogue has no separate add.
//   6 * NK MFMAs of 32 cycles per (32 components x 32 frames x item) instead of KH of 64 cycles:
//   960 vs 2304 cycles for D = 72.
// Image of one item in global memory == its LDS image: [3 splits][NK chunks][64 lanes][8 bf16], i.e.
// every A fragment is one conflict-free, fully contiguous ds_read_b128 per wave.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned fb_pack_hi(float lo, float hi) {  // {lo[31:16], hi[31:16]}
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// round-to-nearest-even to 8 significant bits (a bf16 value held in an f32 register)
__device__ __forceinline__ float fb_bf16_rne(float v) {
  const unsigned u = __float_as_uint(v);
  return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
}
// v[8] -> three packed bf16x8 fragments: v = a + b + c exactly (both residuals are exact in f32 and
// the last one fits 8 bits); rounding to nearest keeps the residuals -- and with them the dropped
// third-order products -- sign-symmetric, so the result carries no systematic bias.
__device__ __forceinline__ void fb_split3_frag(const float (&v)[8], u32x4 &f1, u32x4 &f2, u32x4 &f3) {
  float a[8], b[8], c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = fb_bf16_rne(v[i]);
    const float r = __fsub_rn(v[i], a[i]);
    b[i] = fb_bf16_rne(r);
    c[i] = __fsub_rn(r, b[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f1[i] = fb_pack_hi(a[2 * i], a[2 * i + 1]);
    f2[i] = fb_pack_hi(b[2 * i], b[2 * i + 1]);
    f3[i] = fb_pack_hi(c[2 * i], c[2 * i + 1]);
  }
}

#define FB_BX_MFMA(A, B, ACC) \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)

template <int NK>
__device__ __forceinline__ f32x16 fb_bx_item(const u32x4 *__restrict__ cur4, int lane, const u32x4 (&b1)[NK],
                                             const u32x4 (&b2)[NK], const u32x4 (&b3)[NK], f32x16 acc) {
#pragma unroll
  for (int c = 0; c < NK; ++c) {
    const u32x4 a1 = cur4[(0 * NK + c) * 64 + lane];
    const u32x4 a2 = cur4[(1 * NK + c) * 64 + lane];
    const u32x4 a3 = cur4[(2 * NK + c) * 64 + lane];
    FB_BX_MFMA(a3, b1[c], acc);
    FB_BX_MFMA(a1, b3[c], acc);
    FB_BX_MFMA(a2, b2[c], acc);
    FB_BX_MFMA(a2, b1[c], acc);
    FB_BX_MFMA(a1, b2[c], acc);
    FB_BX_MFMA(a1, b1[c], acc);
  }
  return acc;
}

#ifndef FB_BX_OCC
#define FB_BX_OCC 2
#endif
template <int NK, bool DUMP>
__global__ __launch_bounds__(256, FB_BX_OCC) void k_gmm_bx3(FbGmmDev g, const float *__restrict__ feats,
                                                    const int *__restrict__ n_rows_ptr, int tiles_per_chunk,
                                                    int rows_cap, float *__restrict__ part_m,
                                                    float *__restrict__ part_s, int xcd_map) {
  if (g.stop && *g.stop) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG4 = 3 * NK * 64;  // 16-byte units per item
  constexpr int NST = (IMG4 + 255) / 256;
  const int n_rows = *n_rows_ptr;
  // XCD-aware tile mapping: workgroups go round-robin to the 8 XCDs by linear id; all workgroups of one
  // component chunk are sent to the same XCD(s), so each XCD's L2 holds 1/n_chunks of the parameter images
  // instead of all of them (xcd_map: n_chunks divides 8, see the launcher).
  int strip_i, chunk_i;
  if (xcd_map) {
    const int lin = blockIdx.x, nch = xcd_map, per = 8 / nch;   // XCDs per chunk
    const int xcd = lin & 7, idx = lin >> 3;
    chunk_i = xcd / per;
    strip_i = idx * per + (xcd % per);
  } else {
    strip_i = blockIdx.x;
    chunk_i = blockIdx.y;
  }
  const int strip0 = strip_i * 128;
  if (strip0 >= n_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int row = strip0 + w * 32 + j;
  u32x4 *slot0 = reinterpret_cast<u32x4 *>(lds), *slot1 = slot0 + IMG4;
  float *st_m = lds + 2 * IMG4 * 4;        // [M][256]
  float *st_s = st_m + (size_t)g.M * 256;  // [M][256]

  // ---- frame fragments: chunk c of this lane = dims 16c + 8h + i, i < 8, split into 3 bf16 terms;
  //      bx = x (with 1.0 at the three gconst positions D..D+2), bq = fl(x*x)
  u32x4 bx1[NK], bx2[NK], bx3[NK], bq1[NK], bq2[NK], bq3[NK];
  {
    const bool ok = row < n_rows;
    const float *fr = feats + (size_t)(ok ? row : 0) * g.D;
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      const int d0 = 16 * c + 8 * h;
      float v[8], q[8];
      if ((g.D & 3) == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d = d0 + 4 * u;
          const float4 t = *reinterpret_cast<const float4 *>(fr + min(d, g.D - 4));
          const bool in = ok && d < g.D;
          v[4 * u + 0] = in ? t.x : 0.0f; v[4 * u + 1] = in ? t.y : 0.0f;
          v[4 * u + 2] = in ? t.z : 0.0f; v[4 * u + 3] = in ? t.w : 0.0f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (ok && d0 + i < g.D) ? fr[min(d0 + i, g.D - 1)] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = __fmul_rn(v[i], v[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int d = d0 + i;
        if (d >= g.D && d < g.D + 3) v[i] = 1.0f;
      }
      fb_split3_frag(v, bx1[c], bx2[c], bx3[c]);
      fb_split3_frag(q, bq1[c], bq2[c], bq3[c]);
    }
  }
  for (int m = 0; m < g.M; ++m) { st_m[m * 256 + tid] = FB_GMM_NEG; st_s[m * 256 + tid] = 0.0f; }

  const int tile0 = chunk_i * tiles_per_chunk;
  const int tile1 = min(g.n_tiles, tile0 + tiles_per_chunk);
  const int total_items = (tile1 - tile0) * g.n_items;
  const u32x4 *gimg = g.images_bx + (size_t)tile0 * g.n_items * IMG4;

  u32x4 stage[NST];
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    const int q = min(tid + 256 * s, IMG4 - 1);
    slot0[q] = gimg[q];
  }
  __syncthreads();

  f32x16 accq;
#pragma unroll
  for (int r = 0; r < 16; ++r) accq[r] = 0.0f;

  for (int it = 0; it < total_items; ++it) {
    u32x4 *cur = (it & 1) ? slot1 : slot0;
    u32x4 *nxt = (it & 1) ? slot0 : slot1;
    {
      const u32x4 *src = gimg + (size_t)min(it + 1, total_items - 1) * IMG4;
#pragma unroll
      for (int s = 0; s < NST; ++s) stage[s] = src[min(tid + 256 * s, IMG4 - 1)];
    }
    const int item = it % g.n_items;
    const int model = g.item_model[item];
    if (model < 0) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      accq = fb_bx_item<NK>(cur, lane, bq1, bq2, bq3, acc);
    } else {
      const f32x16 v = fb_bx_item<NK>(cur, lane, bx1, bx2, bx3, accq);
      if constexpr (DUMP) {
        if (row < n_rows) {
          const int tile = tile0 + it / g.n_items;
          float *dst = part_m + (size_t)row * (g.n_tiles * 32) + tile * 32 + 4 * h;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            *reinterpret_cast<float4 *>(dst + 8 * rr) = make_float4(v[4 * rr], v[4 * rr + 1], v[4 * rr + 2], v[4 * rr + 3]);
        }
      } else {
        float tm = FB_GMM_NEG;
#pragma unroll
        for (int r = 0; r < 16; ++r) tm = fmaxf(tm, v[r]);
        const float m_old = st_m[model * 256 + tid], s_old = st_s[model * 256 + tid];
        const float m_new = fmaxf(m_old, tm);
        float ssum = s_old * __expf(m_old - m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) ssum += __expf(v[r] - m_new);
        st_m[model * 256 + tid] = m_new;
        st_s[model * 256 + tid] = ssum;
      }
    }
#pragma unroll
    for (int s = 0; s < NST; ++s) nxt[min(tid + 256 * s, IMG4 - 1)] = stage[s];
    __syncthreads();
  }

  if constexpr (DUMP) return;
  for (int m = 0; m < g.M; ++m) {
    const float mm = st_m[m * 256 + tid], ss = st_s[m * 256 + tid];
    const float m2 = __shfl_xor(mm, 32, 64), s2 = __shfl_xor(ss, 32, 64);
    const float mx = fmaxf(mm, m2);
    const float sx = ss * __expf(mm - mx) + s2 * __expf(m2 - mx);
    if (h == 0 && row < n_rows) {
      const size_t o = ((size_t)chunk_i * g.M + m) * rows_cap + row;
      part_m[o] = mx;
      part_s[o] = sx;
    }
  }
}

template <int NK, bool DUMP>
static void launch_gmm_bx_t(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                            int rows_cap, int n_chunks, int tpc, float *part_m, float *part_s) {
  const int strips = (rows_cap + 127) / 128;
  dim3 grid((unsigned)strips, (unsigned)n_chunks);
  int xcd_map = 0;
  static const bool no_xcd_map = getenv("FB_GMM_NO_XCD_MAP") != nullptr;
  if ((n_chunks == 1 || n_chunks == 2 || n_chunks == 4 || n_chunks == 8) && !no_xcd_map) {
    const int per = 8 / n_chunks;  // XCDs serving one component chunk
    grid = dim3((unsigned)(8 * ((strips + per - 1) / per)), 1);
    xcd_map = n_chunks;
  }
  const size_t ldsb = (size_t)2 * 3 * NK * 64 * 16 + (size_t)2 * g.M * 256 * sizeof(float);
  hipLaunchKernelGGL((k_gmm_bx3<NK, DUMP>), grid, dim3(256), ldsb, s, g, feats, n_rows_ptr, tpc, rows_cap,
                     part_m, part_s, xcd_map);
}
template <bool DUMP>
static void launch_gmm_bx(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                          int rows_cap, int n_chunks, int tpc, float *part_m, float *part_s) {
  switch (g.NK) {
    case 3: launch_gmm_bx_t<3, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 4: launch_gmm_bx_t<4, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 5: launch_gmm_bx_t<5, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 6: launch_gmm_bx_t<6, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    default: break;  // fb_load_gmm only produces the NK values above
  }
}

// ---------------------------------------------------------------------------------------------
// k_gmm_fx2: the same computation on the f16 matrix pipe with a TWO-way split.
// fl16 keeps 11 significant bits and the residual of a round-to-nearest split carries its own sign,
// so  v = v1 + v2 + r,  v1 = fl16(v),  v2 = fl16(v - v1),  |r| <= 2^-24 |v|:  two f16 terms represent an
// f32 value to within half an f32 ulp.  A product needs
//     a1b1 + (a1b2 + a2b1)          (dropped: a2b2 <= 2^-24 |ab|, the size of one f32 rounding)
// i.e. 3 MFMAs per 16 K instead of bx3's 6, at the same matrix rate.  Measured against float64 the
// result is as close as the f32 MFMA kernel's (numpy model in tools/probes/fx2_emul.py).
// f16's narrow exponent range is handled without data-dependent scaling:
//   * all three products go into ONE accumulator; residuals are stored unscaled.  The f16 matrix pipe keeps
//     subnormal inputs (tools/probes/f16_denorm_probe.hip), so a residual below 2^-14 is still exact to 2^-25 absolute.
//   * operands are moved up by exact powers of two chosen at load time (fb_load_gmm): (mu/sigma^2, gconst) * 2^kl
//     against (x, 1) * 2^kx, and -1/(2 sigma^2) * 2^kq against x^2 * 2^kx2 with kl + kx = kq + kx2 = kacc, so
//     that typical residuals are normal numbers and the largest operand stays below 2^15 (|x| < 4094, |x| < 511
//     for the squares).  The accumulators hold ll * 2^kacc; the logsumexp update folds 2^-kacc into the
//     multiplier of its fma, so the scaling costs nothing.
//   * fb_load_gmm selects this kernel only if every parameter fits; otherwise bx3 runs.
//   (An earlier form kept the residuals * 2^12 in a second accumulator, value = hi + 2^-12 mid: 135 us instead of
//    129 us and 32 more VGPRs.)
// Image of one item: [2 terms][NK chunks][64 lanes][8 f16]; gconst sits at K position D (its two terms
// against 2^kx / 0 in the frame operand).
// One item of the k_gmm_fx2 loop: the 3*NK MFMAs of a 32-component x 32-frame tile.
//   ISQ : quadratic item: the accumulator starts from zero and is kept in hq for the models that follow
//   else: model item: continues from hq; the values (ll * 2^kacc) are left in pv
// (Tried and measured slower, DESIGN.md §5: interleaving the previous item's logsumexp update into this MFMA chain with
// sched_group_barrier, s_setprio around the chain, A fragments fetched one K step ahead to fit 3 waves per SIMD.)
template <int NK, bool ISQ>
__device__ __forceinline__ void fb_fx_step(const u32x4 *__restrict__ cur4, int lane, const u32x4 (&b1)[NK],
                                           const u32x4 (&b2)[NK], f32x16 &hq, f32x16 &pv) {
  f32x16 hi;
  if constexpr (ISQ) {
#pragma unroll
    for (int r = 0; r < 16; ++r) hi[r] = 0.0f;
  } else {
    hi = hq;
  }
  u32x4 a1[NK], a2[NK];
#pragma unroll
  for (int c = 0; c < NK; ++c) {
    a1[c] = cur4[(0 * NK + c) * 64 + lane];
    a2[c] = cur4[(1 * NK + c) * 64 + lane];
  }
#pragma unroll
  for (int c = 0; c < NK; ++c) {
    FB_FX_MFMA(a2[c], b1[c], hi);
    FB_FX_MFMA(a1[c], b2[c], hi);
    FB_FX_MFMA(a1[c], b1[c], hi);
  }
  if constexpr (ISQ) hq = hi; else pv = hi;
}

// (fb_fx_frame_frags -- the frame operands of a lane -- lives in gmm_split.h: k_gsel_w of gmm_wide_kernel.hip shares it)
template <int NK, bool DUMP>
__global__ __launch_bounds__(256, FB_FX_OCC) void k_gmm_fx2(FbGmmDev g, const float *__restrict__ feats,
                                                    const int *__restrict__ n_rows_ptr, int tiles_per_chunk,
                                                    int rows_cap, float *__restrict__ part_m,
                                                    float *__restrict__ part_s, int xcd_map) {
  if (g.stop && *g.stop) return;
  if constexpr (DUMP) {
    if (g.only_if && *g.only_if == 0) return;  // the gselect rescue: nothing overflowed
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG4 = 2 * NK * 64;  // 16-byte units per item
  constexpr int NST = (IMG4 + 255) / 256;
  const int n_rows = *n_rows_ptr;
  int strip_i, chunk_i;  // XCD-aware (strip, chunk) mapping, as in k_gmm_bx3
  if (xcd_map) {
    const int lin = blockIdx.x, per = 8 / xcd_map;
    const int xcd = lin & 7, idx = lin >> 3;
    chunk_i = xcd / per;
    strip_i = idx * per + (xcd % per);
  } else {
    strip_i = blockIdx.x;
    chunk_i = blockIdx.y;
  }
  const int strip0 = strip_i * 128;
  if (strip0 >= n_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int row = strip0 + w * 32 + j;
  u32x4 *slot0 = reinterpret_cast<u32x4 *>(lds), *slot1 = slot0 + IMG4;
  float *st_m = lds + 2 * IMG4 * 4;        // [M][256]
  float *st_s = st_m + (size_t)g.M * 256;  // [M][256]

  // ---- frame fragments: chunk c of this lane = dims 16c + 8h + i, i < 8;  bx = x (1.0 at position D,
  //      whose residual is 0), bq = fl(x*x), both moved by the load-time powers of two 2^kx / 2^kx2.
  // Range guard: the load-time scalings assume |x| 2^kx and x^2 2^kx2 below f16's 65504.  Features beyond
  // that (|x| >= 64 with kx2 = 4: liftered cepstra of tonal audio, unusual front-end configurations) would turn
  // into inf and the scores into NaN.  Each wave therefore takes the largest scaled operand of its 32 frames
  // and, when it reaches 2^15, moves ALL its frame operands (x, the 1.0 that multiplies gconst, x^2) down by one
  // wave-uniform power of two 2^-sh: the accumulators then hold ll 2^(kacc - sh), the logsumexp multiplier and
  // the final un-scaling take the factor back, and every step stays an exact power-of-two scaling.  Small
  // operands of such a frame may become f16 subnormals (absolute precision 2^-25 of the scaled operand), which is
  // below the f32 rounding of the large terms that caused the shift.  sh = 0 for ordinary speech features.
  u32x4 bx1[NK], bx2[NK], bq1[NK], bq2[NK];
  const int sh = fb_fx_frame_frags<NK>(g, feats, row, n_rows, h, bx1, bx2, bq1, bq2);
  for (int m = 0; m < g.M; ++m) { st_m[m * 256 + tid] = FB_GMM_NEG; st_s[m * 256 + tid] = 0.0f; }

  const int tile0 = chunk_i * tiles_per_chunk;
  const int tile1 = min(g.n_tiles, tile0 + tiles_per_chunk);
  const int total_items = (tile1 - tile0) * g.n_items;
  const u32x4 *gimg = g.images_fx + (size_t)tile0 * g.n_items * IMG4;

  u32x4 stage[NST];
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    const int q = min(tid + 256 * s, IMG4 - 1);
    slot0[q] = gimg[q];
  }
  __syncthreads();

  const float unscale = fb_pow2f(sh - g.kacc), ls = __fmul_rn(FB_LOG2E_F, unscale);  // exact: a power of two
  const int pad_it0 = ((g.C & 31) && tile1 == g.n_tiles) ? (tile1 - 1 - tile0) * g.n_items : 0x7fffffff;
  f32x16 hq, pv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { hq[r] = 0.0f; pv[r] = 0.0f; }

  for (int it = 0; it < total_items; ++it) {
    u32x4 *cur = (it & 1) ? slot1 : slot0;
    u32x4 *nxt = (it & 1) ? slot0 : slot1;
    {
      const u32x4 *src = gimg + (size_t)min(it + 1, total_items - 1) * IMG4;
#pragma unroll
      for (int s = 0; s < NST; ++s) stage[s] = src[min(tid + 256 * s, IMG4 - 1)];
    }
    const int item = it % g.n_items;
    const int model = g.item_model[item];
    if (model < 0) {
      fb_fx_step<NK, true>(cur, lane, bq1, bq2, hq, pv);
    } else {
      fb_fx_step<NK, false>(cur, lane, bx1, bx2, hq, pv);
      if constexpr (DUMP) {
        // The wave's 32 frames x 32 components go out as WHOLE 128-byte rows: through a per-wave LDS tile (row stride 36
        // floats) into the order "eight consecutive lanes = one frame's row", four store instructions of eight full
        // lines each.  Straight from the accumulator layout a lane held four separate 16-byte pieces of its frame's
        // row and every store instruction touched 32 lines with 32 bytes each: the PMC pass counted 234 MB written for
        // 125 MB of values (profiles/r04_traffic.json).
        const int tile = tile0 + it / g.n_items;
        float *tb = lds + 2 * IMG4 * 4 + 2 * g.M * 256 + w * (32 * 36);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          *reinterpret_cast<float4 *>(tb + j * 36 + 8 * rr + 4 * h) = make_float4(pv[4 * rr] * unscale, pv[4 * rr + 1] * unscale,
                                                                              pv[4 * rr + 2] * unscale, pv[4 * rr + 3] * unscale);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int fr = 8 * i + (lane >> 3), piece = lane & 7, rg = strip0 + w * 32 + fr;
          const float4 v = *reinterpret_cast<const float4 *>(tb + fr * 36 + 4 * piece);
          if (rg < n_rows) *reinterpret_cast<float4 *>(part_m + (size_t)rg * (g.n_tiles * 32) + tile * 32 + 4 * piece) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the tile is read before the next item overwrites it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else {
        if (it >= pad_it0) {  // last tile of a model whose C is not a multiple of 32: the padding components' gconst
                              // (-60000 2^-kl, the most an f16 image can hold) must not compete with far-off frames
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if ((g.n_tiles - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= g.C) pv[r] = FB_GMM_NEG;
        }
        fb_lse_update16(pv, st_m + model * 256 + tid, st_s + model * 256 + tid, ls);
      }
    }
#pragma unroll
    for (int s = 0; s < NST; ++s) nxt[min(tid + 256 * s, IMG4 - 1)] = stage[s];
    __syncthreads();
  }

  if constexpr (DUMP) return;
  for (int m = 0; m < g.M; ++m) {
    const float ms = st_m[m * 256 + tid];  // maximum of ll * 2^kacc
    const float mm = ms * unscale, ss = fb_lse_to_natural(ms, st_s[m * 256 + tid], ls);
    const float m2 = __shfl_xor(mm, 32, 64), s2 = __shfl_xor(ss, 32, 64);
    const float mx = fmaxf(mm, m2);
    const float sx = ss * __expf(mm - mx) + s2 * __expf(m2 - mx);
    if (h == 0 && row < n_rows) {
      const size_t o = ((size_t)chunk_i * g.M + m) * rows_cap + row;
      part_m[o] = mx;
      part_s[o] = sx;
    }
  }
}


template <int NK, bool DUMP>
static void launch_gmm_fx_t(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                            int rows_cap, int n_chunks, int tpc, float *part_m, float *part_s) {
  const int strips = (rows_cap + 127) / 128;
  dim3 grid((unsigned)strips, (unsigned)n_chunks);
  int xcd_map = 0;
  static const bool no_xcd_map = getenv("FB_GMM_NO_XCD_MAP") != nullptr;
  if ((n_chunks == 1 || n_chunks == 2 || n_chunks == 4 || n_chunks == 8) && !no_xcd_map) {
    const int per = 8 / n_chunks;
    grid = dim3((unsigned)(8 * ((strips + per - 1) / per)), 1);
    xcd_map = n_chunks;
  }
  const size_t ldsb = (size_t)2 * 2 * NK * 64 * 16 + (size_t)2 * g.M * 256 * sizeof(float) +
                      (DUMP ? (size_t)4 * 32 * 36 * sizeof(float) : 0);   // + the dump's per-wave transposition tiles
  hipLaunchKernelGGL((k_gmm_fx2<NK, DUMP>), grid, dim3(256), ldsb, s, g, feats, n_rows_ptr, tpc, rows_cap,
                     part_m, part_s, xcd_map);
}
template <bool DUMP>
static void launch_gmm_fx(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                          int rows_cap, int n_chunks, int tpc, float *part_m, float *part_s) {
  switch (g.NKF) {
    case 2: launch_gmm_fx_t<2, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 3: launch_gmm_fx_t<3, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 4: launch_gmm_fx_t<4, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 5: launch_gmm_fx_t<5, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 6: launch_gmm_fx_t<6, DUMP>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    default: break;  // fb_load_gmm only produces the NKF values above
  }
}


// ------------------------------------------------------------------------------------------------
// gmm-gselect --n=20 WITHOUT the dump (round 6; ivector_PLDA_kaldiHelper.py:197-213).
// The dump + k_iv_select pair wrote and re-read every one of the C log-likelihoods of every frame (2 x 125 MB at
// configs[2] size for 1.2 MB of indices).  Here nothing but survivors of a threshold leaves the matrix-core kernel:
//   pass A  k_gmm_fx2_sel<NK, false>: the k_gmm_fx2 body; a lane (one frame, 16 components of the tile) keeps the MAXIMUM
//           of its 16 values -> gmax[row][2 n_tiles] (one float per (tile, half): 128 per frame at C = 2048).
//   tau     k_gsel_tau: tau(row) = the nsel-th largest of the row's group maxima.  nsel distinct groups hold a value
//           >= tau, so the nsel-th largest log-likelihood is >= tau: NO component below tau can be selected.
//   pass B  k_gmm_fx2_sel<NK, true>: the same body again (bit-identical values); every value >= tau(row) is appended as a
//           64-bit key (ordered value bits, component index) to the row's list of this component chunk (LDS, then global).
//   final   k_gsel_final: the row's survivors (~22 of 2048) ranked by counting (key > key) -- descending (value, index),
//           std::greater<pair<float,int>> like gmm-gselect and k_iv_select -- ranks < nsel written to sel[].
// Exact by construction: the selection is a function of the same float values the dump would have stored.  A list that
// overflows its fb_gsel_cap() entries (never seen on speech; possible with degenerate models) raises `flag`, and the caller's
// dump + k_iv_select launches -- which return at once while the flag is zero -- redo the batch the old way.
#define FB_GSEL_MAXC 128   // survivors per row the final kernel ranks (n_chunks x cap)
int fb_gsel_cap(int n_chunks) {
  const char *ev = getenv("FB_GSEL_CAP");   // tests: a tiny capacity makes every list overflow (the rescue path)
  if (ev && atoi(ev) > 0 && atoi(ev) <= 32) return atoi(ev);
  return n_chunks <= 4 ? 32 : (n_chunks <= 8 ? 16 : 0);
}
// component chunks of the selection kernels: the dump's count brought down to a power of two <= 8 (the lists of a row
// are n_chunks x cap <= FB_GSEL_MAXC entries)
int fb_gsel_chunks(int n_chunks) { return n_chunks >= 8 ? 8 : (n_chunks >= 4 ? 4 : (n_chunks >= 2 ? 2 : 1)); }
bool fb_gsel_applies(const FbGmmDev &g, int nsel, int n_chunks) {
  // (few groups: tau would be -inf and every component a survivor -- the dump is the right tool for small models)
  const bool off = getenv("FB_IV_GSEL_DUMP") != nullptr;   // A/B and tests: the dump + k_iv_select path (read per batch)
  return !off && g.mode == FB_GMM_MODE_FX2 && g.M == 1 && g.n_items == 2 && 2 * g.n_tiles >= 4 * nsel && 2 * g.n_tiles <= 256 &&
         fb_gsel_cap(n_chunks) > 0 && nsel <= 32;
}
__device__ __forceinline__ unsigned fb_f32_ordered(float v) {  // monotone map float -> unsigned (total order of the values)
  const unsigned u = __float_as_uint(v);
  return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}
template <int NK, bool PICK>
__global__ __launch_bounds__(256, FB_FX_OCC) void k_gmm_fx2_sel(FbGmmDev g, const float *__restrict__ feats,
                                                        const int *__restrict__ n_rows_ptr, int tiles_per_chunk, int n_chunks,
                                                        float *__restrict__ gmax, const float *__restrict__ tau, int cap,
                                                        unsigned long long *__restrict__ glist, int *__restrict__ gcnt,
                                                        int xcd_map) {
  if (g.stop && *g.stop) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG4 = 2 * NK * 64;  // 16-byte units per item
  constexpr int NST = (IMG4 + 255) / 256;
  const int n_rows = *n_rows_ptr;
  int strip_i, chunk_i;  // XCD-aware (strip, chunk) mapping, as in k_gmm_bx3
  if (xcd_map) {
    const int lin = blockIdx.x, per = 8 / xcd_map;
    const int xcd = lin & 7, idx = lin >> 3;
    chunk_i = xcd / per;
    strip_i = idx * per + (xcd % per);
  } else {
    strip_i = blockIdx.x;
    chunk_i = blockIdx.y;
  }
  const int strip0 = strip_i * 128;
  if (strip0 >= n_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int row = strip0 + w * 32 + j;
  u32x4 *slot0 = reinterpret_cast<u32x4 *>(lds), *slot1 = slot0 + IMG4;
  float *s_mx = lds + 2 * IMG4 * 4;                                                   // pass A: [tiles_per_chunk][256]
  int *s_cnt = reinterpret_cast<int *>(lds + 2 * IMG4 * 4);                           // pass B: [128]
  unsigned long long *s_list = reinterpret_cast<unsigned long long *>(s_cnt + 128);  //         [128][cap]
  u32x4 bx1[NK], bx2[NK], bq1[NK], bq2[NK];
  const int sh = fb_fx_frame_frags<NK>(g, feats, row, n_rows, h, bx1, bx2, bq1, bq2);
  float my_tau = FLT_MAX;
  if constexpr (PICK) {
    if (row < n_rows) my_tau = tau[row];
    if (tid < 128) s_cnt[tid] = 0;
  }
  const int tile0 = chunk_i * tiles_per_chunk;
  const int tile1 = min(g.n_tiles, tile0 + tiles_per_chunk);
  const int total_items = (tile1 - tile0) * 2;   // {quadratic item, the model's item} per tile
  const u32x4 *gimg = g.images_fx + (size_t)tile0 * 2 * IMG4;
  u32x4 stage[NST];
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    const int q = min(tid + 256 * s, IMG4 - 1);
    slot0[q] = gimg[q];
  }
  __syncthreads();
  const float unscale = fb_pow2f(sh - g.kacc);
  f32x16 hq, pv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { hq[r] = 0.0f; pv[r] = 0.0f; }
  const int rl = w * 32 + j;   // the frame's row inside the strip
  for (int it = 0; it < total_items; ++it) {
    u32x4 *cur = (it & 1) ? slot1 : slot0;
    u32x4 *nxt = (it & 1) ? slot0 : slot1;
    {
      const u32x4 *src = gimg + (size_t)min(it + 1, total_items - 1) * IMG4;
#pragma unroll
      for (int s = 0; s < NST; ++s) stage[s] = src[min(tid + 256 * s, IMG4 - 1)];
    }
    if (!(it & 1)) {
      fb_fx_step<NK, true>(cur, lane, bq1, bq2, hq, pv);
    } else {
      fb_fx_step<NK, false>(cur, lane, bx1, bx2, hq, pv);
      const int tl = it >> 1, cbase = (tile0 + tl) * 32 + 4 * h;   // accumulator r = component cbase + (r & 3) + 8 (r >> 2)
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = pv[r] * unscale;   // the float the dump stores (exact: a power of two)
      if (cbase + 27 >= g.C) {   // the last tile of a model whose C is not a multiple of 32: padding never competes
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (cbase + (r & 3) + 8 * (r >> 2) >= g.C) v[r] = -FLT_MAX;
      }
      if constexpr (!PICK) {
        float tm = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, v[r]);
        s_mx[tl * 256 + tid] = tm;
      } else {
        int n = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) n += (v[r] >= my_tau) ? 1 : 0;
        if (n > 0) {
          int at = atomicAdd(&s_cnt[rl], n);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (v[r] >= my_tau) {
              if (at < cap)
                s_list[rl * cap + at] = ((unsigned long long)fb_f32_ordered(v[r]) << 32) | (unsigned)(cbase + (r & 3) + 8 * (r >> 2));
              ++at;
            }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NST; ++s) nxt[min(tid + 256 * s, IMG4 - 1)] = stage[s];
    __syncthreads();
  }
  const int ntl = tile1 - tile0;
  if constexpr (!PICK) {
    // whole 8 ntl-byte runs per row: consecutive threads = consecutive (tile, half) positions of one frame
    const int NG = 2 * g.n_tiles, per_row = 2 * ntl;
    for (int i = tid; i < 128 * per_row; i += 256) {
      const int r2 = i / per_row, pos = i - r2 * per_row;
      const int tl = pos >> 1, hh = pos & 1;
      const int rg = strip0 + r2;
      if (rg < n_rows) gmax[(size_t)rg * NG + 2 * tile0 + pos] = s_mx[tl * 256 + (r2 >> 5) * 64 + hh * 32 + (r2 & 31)];
    }
  } else {
    (void)ntl;
    for (int i = tid; i < 128 * cap; i += 256) {
      const int r2 = i / cap, k = i - r2 * cap;
      const int rg = strip0 + r2;
      if (rg < n_rows && k < min(s_cnt[r2], cap)) glist[((size_t)rg * n_chunks + chunk_i) * cap + k] = s_list[i];
    }
    if (tid < 128 && strip0 + tid < n_rows) gcnt[(size_t)(strip0 + tid) * n_chunks + chunk_i] = s_cnt[tid];
  }
}

// tau(row) = the nsel-th largest of the row's NG = 16 NV group maxima.  A frame per DPP row (16 lanes, NV values each in
// registers), four frames per wave; a round = the row maximum (four DPP steps) and its removal.  Equal maxima leave
// together: after nsel rounds at least nsel groups >= the last maximum have left -- all tau has to promise.
template <int NV>
__global__ __launch_bounds__(256) void k_gsel_tau(const float *__restrict__ gmax, int NG, const int *__restrict__ n_rows_ptr,
                                                 int nsel, float *__restrict__ tau, int *__restrict__ flag) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 0;   // (the rescue launches of the previous batch are behind us)
  const int n_rows = *n_rows_ptr;
  const int l = threadIdx.x & 15, row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool ok = row < n_rows;
  const float *gr = gmax + (size_t)(ok ? row : 0) * NG;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (ok && l + 16 * i < NG) ? gr[min(l + 16 * i, NG - 1)] : -FLT_MAX;
  float t = -FLT_MAX;
  for (int s = 0; s < nsel; ++s) {
    float m = v[0];
#pragma unroll
    for (int i = 1; i < NV; ++i) m = fmaxf(m, v[i]);
#define FB_RMAX(CTRL) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), CTRL, 0xf, 0xf, false)));
    FB_RMAX(0xb1) FB_RMAX(0x4e) FB_RMAX(0x141) FB_RMAX(0x140)
#undef FB_RMAX
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = v[i] >= m ? -FLT_MAX : v[i];
    t = m;
  }
  if (ok && l == 0) tau[row] = t;
}

// the survivors of a row ranked: 16 lanes per frame, 16 frames per workgroup.  rank(key) = number of larger keys; the keys
// are distinct (the index is part of them), so ranks 0 .. n-1 are a permutation: descending (value, index).
__global__ __launch_bounds__(256) void k_gsel_final(const unsigned long long *__restrict__ glist, const int *__restrict__ gcnt,
                                                   int n_chunks, int cap, const int *__restrict__ n_rows_ptr, int nsel, int C,
                                                   int *__restrict__ sel, int *__restrict__ flag) {
  __shared__ unsigned long long s_key[16][FB_GSEL_MAXC];
  const int n_rows = *n_rows_ptr;
  const int l = threadIdx.x & 15, rw = threadIdx.x >> 4, row = blockIdx.x * 16 + rw;
  if (row >= n_rows) return;   // (no workgroup barrier below: a frame's 16 lanes sit in one wave)
  int n = 0;
  bool over = false;
  for (int k = 0; k < n_chunks; ++k) {
    const int c = gcnt[(size_t)row * n_chunks + k];
    over |= c > cap;
    const int cc = min(c, cap);
    const unsigned long long *src = glist + ((size_t)row * n_chunks + k) * cap;
    for (int e = l; e < cc; e += 16) s_key[rw][n + e] = src[e];
    n += cc;
  }
  if (over || n < min(nsel, C)) { if (l == 0) atomicOr(flag, 1); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int i = l; i < n; i += 16) {
    const unsigned long long my = s_key[rw][i];
    int rank = 0;
    for (int q = 0; q < n; ++q) rank += s_key[rw][q] > my ? 1 : 0;
    if (rank < nsel) sel[(size_t)row * nsel + rank] = (int)(unsigned)(my & 0xffffffffull);
  }
  for (int s2 = n + l; s2 < nsel; s2 += 16) sel[(size_t)row * nsel + s2] = -1;   // fewer components than nsel
}

template <int NK>
static void launch_gsel_t(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap, int n_chunks,
                          int nsel, float *gmax, float *tau, unsigned long long *glist, int *gcnt, int *flag, int *sel) {
  const int strips = (rows_cap + 127) / 128, tpc = (g.n_tiles + n_chunks - 1) / n_chunks, cap = fb_gsel_cap(n_chunks);
  dim3 grid((unsigned)strips, (unsigned)n_chunks);
  int xcd_map = 0;
  static const bool no_xcd_map = getenv("FB_GMM_NO_XCD_MAP") != nullptr;
  if ((n_chunks == 1 || n_chunks == 2 || n_chunks == 4 || n_chunks == 8) && !no_xcd_map) {
    const int per = 8 / n_chunks;
    grid = dim3((unsigned)(8 * ((strips + per - 1) / per)), 1);
    xcd_map = n_chunks;
  }
  const size_t img = (size_t)2 * 2 * NK * 64 * 16;
  hipLaunchKernelGGL((k_gmm_fx2_sel<NK, false>), grid, dim3(256), img + sizeof(float) * 256 * (size_t)tpc, s, g, feats, n_rows_ptr, tpc,
                     n_chunks, gmax, nullptr, cap, nullptr, nullptr, xcd_map);
  const int NG = 2 * g.n_tiles, tb = (rows_cap + 15) / 16;
  if (NG <= 64) hipLaunchKernelGGL(k_gsel_tau<4>, dim3(tb), dim3(256), 0, s, gmax, NG, n_rows_ptr, nsel, tau, flag);
  else if (NG <= 128) hipLaunchKernelGGL(k_gsel_tau<8>, dim3(tb), dim3(256), 0, s, gmax, NG, n_rows_ptr, nsel, tau, flag);
  else hipLaunchKernelGGL(k_gsel_tau<16>, dim3(tb), dim3(256), 0, s, gmax, NG, n_rows_ptr, nsel, tau, flag);
  hipLaunchKernelGGL((k_gmm_fx2_sel<NK, true>), grid, dim3(256), img + sizeof(int) * 128 + sizeof(unsigned long long) * 128 * (size_t)cap, s, g,
                     feats, n_rows_ptr, tpc, n_chunks, nullptr, tau, cap, glist, gcnt, xcd_map);
  hipLaunchKernelGGL(k_gsel_final, dim3(tb), dim3(256), 0, s, glist, gcnt, n_chunks, cap, n_rows_ptr, nsel, g.C, sel, flag);
}
void fb_launch_gsel(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap, int n_chunks,
                    int nsel, float *gmax, float *tau, unsigned long long *glist, int *gcnt, int *flag, int *sel) {
  if (rows_cap <= 0) return;
  switch (g.NKF) {
    case 2: launch_gsel_t<2>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, nsel, gmax, tau, glist, gcnt, flag, sel); break;
    case 3: launch_gsel_t<3>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, nsel, gmax, tau, glist, gcnt, flag, sel); break;
    case 4: launch_gsel_t<4>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, nsel, gmax, tau, glist, gcnt, flag, sel); break;
    case 5: launch_gsel_t<5>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, nsel, gmax, tau, glist, gcnt, flag, sel); break;
    case 6: launch_gsel_t<6>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, nsel, gmax, tau, glist, gcnt, flag, sel); break;
    default: break;
  }
}


// ---- the wide form (k_gsel_w, gmm_wide_kernel.hip): pass A -> k_gsel_tau -> pass B -> k_gsel_final_w.  Pass B leaves, per
// (row, chunk), the 16-value records of the groups whose maximum reaches tau -- every group has a place, nothing overflows,
// so no rescue launches follow.  The final kernel ranks the values >= tau of a row's records (16 lanes per frame, 16 frames
// per workgroup); the keys are (ordered value bits, component index) as in k_gsel_final, in the accumulators' own scale.
bool fb_gsel_w_applies(const FbGmmDev &g, int n_chunks);   // gmm_wide_kernel.hip
void fb_launch_gsel_w(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap, int n_chunks, int pick,
                      float *gmax, const float *tau, float *gval, unsigned char *gid, int *gcnt, int tiles_a);
int fb_gsel_w_tiles_a(const FbGmmDev &g, int n_chunks, int nsel);
int fb_gsel_wide_chunks(const FbGmmDev &g, int nsel, int rows_cap, int target_blocks) {
  if (getenv("FB_IV_GSEL_DUMP") != nullptr) return 0;
  if (!(g.mode == FB_GMM_MODE_FX2 && g.M == 1 && g.n_items == 2 && 2 * g.n_tiles >= 4 * nsel && 2 * g.n_tiles <= 256 && nsel <= 32)) return 0;
  const int strips = (rows_cap + 255) / 256;
  if (const char *ev = getenv("FB_GSEL_TARGET_BLOCKS")) target_blocks = atoi(ev);
  int want = (target_blocks > 0 ? target_blocks : 256) / (strips > 0 ? strips : 1);
  if (want < 1) want = 1;
  int n = 8;
  while (n > 1 && (n > want || !fb_gsel_w_applies(g, n))) n >>= 1;
  return fb_gsel_w_applies(g, n) ? n : 0;
}
// A workgroup = FB_GSEL_FB = 64 frames (1024 threads) = one partition block of the bucket sort that follows: the counts
// k_iv_bucket_count would make of sel[] (ivector_kernels.hip: per-component counts of the block) are taken here from the
// selections as they are written -- cnt[block][Cpad], nullable -- and that launch is not made.
#define FB_GSEL_FB FB_IV_FB
#define FB_GSEL_MAXW 64   // values at or above tau a frame ranks through LDS (more: straight from the records)
__global__ __launch_bounds__(1024) void k_gsel_final_w(const float *__restrict__ gval, const unsigned char *__restrict__ gid,
                                                      const int *__restrict__ gcnt, const float *__restrict__ tau, int n_chunks, int tpc,
                                                      const int *__restrict__ n_rows_ptr, int nsel, int C, int *__restrict__ sel,
                                                      int *__restrict__ flag, int *__restrict__ cnt, int Cpad) {
  __shared__ unsigned long long s_key[FB_GSEL_FB][FB_GSEL_MAXW];
  __shared__ int s_n[FB_GSEL_FB];
  extern __shared__ int s_hist[];   // [Cpad] when cnt
  if (cnt) {
    for (int i = threadIdx.x; i < Cpad; i += 1024) s_hist[i] = 0;
    __syncthreads();
  }
  const int n_rows = *n_rows_ptr;
  const int l = threadIdx.x & 15, rw = threadIdx.x >> 4, row = blockIdx.x * FB_GSEL_FB + rw;
  if (row < n_rows) {   // (no workgroup barrier inside: a frame's 16 lanes sit in one wave)
  const int capc = 2 * tpc;
  if (l == 0) s_n[rw] = 0;
  const float t = tau[row];
  int pre[9];   // records of the chunks before chunk k (n_chunks <= 8)
  pre[0] = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) pre[k + 1] = pre[k] + (k < n_chunks ? min(gcnt[(size_t)row * n_chunks + k], capc) : 0);
  const int total = pre[8];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // record e of the row: (chunk, index in the chunk) -> its 16 values and the component of value 0
  auto record = [&](int e, const float4 *&src, int &cbase) {
    int k = 0, first = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q)
      if (e >= pre[q]) { k = q; first = pre[q]; }
    const size_t at = ((size_t)row * n_chunks + k) * capc + (e - first);
    const int gl = gid[at];
    src = reinterpret_cast<const float4 *>(gval) + at * 4;
    cbase = (k * tpc + (gl >> 1)) * 32 + 4 * (gl & 1);   // value r = component cbase + (r & 3) + 8 (r >> 2)
  };
  // four records per lane at a time: their ids first, then all their values in flight together; a lane reserves the places of
  // its values >= tau with ONE addition to the row's counter
  for (int e0 = 0; e0 < total; e0 += 64) {
    const float4 *src[4]; int cbase[4]; bool have[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + l + 16 * q;
      have[q] = e < total;
      record(have[q] ? e : 0, src[q], cbase[q]);
    }
    float4 f[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) f[q][u] = src[q][u];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        mine += have[q] ? ((f[q][u].x >= t ? 1 : 0) + (f[q][u].y >= t ? 1 : 0) + (f[q][u].z >= t ? 1 : 0) + (f[q][u].w >= t ? 1 : 0)) : 0;
    int pos = mine > 0 ? atomicAdd(&s_n[rw], mine) : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v4[4] = {f[q][u].x, f[q][u].y, f[q][u].z, f[q][u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (have[q] && v4[i] >= t) {   // value r = 4 u + i = component cbase + (r & 3) + 8 (r >> 2)
            if (pos < FB_GSEL_MAXW) s_key[rw][pos] = ((unsigned long long)fb_f32_ordered(v4[i]) << 32) | (unsigned)(cbase[q] + i + 8 * u);
            ++pos;
          }
      }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int n = s_n[rw];
  if (n < min(nsel, C)) { if (l == 0) atomicOr(flag, 1); }   // (NaN features only: tau is the nsel-th largest group maximum)
  if (n <= FB_GSEL_MAXW) {
    for (int i = l; i < n; i += 16) {
      const unsigned long long my = s_key[rw][i];
      int rank = 0;
      for (int q = 0; q < n; ++q) rank += s_key[rw][q] > my ? 1 : 0;
      if (rank < nsel) {
        const int comp = (int)(unsigned)(my & 0xffffffffull);
        sel[(size_t)row * nsel + rank] = comp;
        if (cnt) atomicAdd(&s_hist[comp], 1);
      }
    }
  } else {
    // more values at or above tau than the key list holds (ties by the hundred: degenerate models, constant features): rank
    // straight from the records -- quadratic in their number, never taken on speech
    for (int i = l; i < total * 16; i += 16) {
      const float4 *src; int cbase;
      record(i >> 4, src, cbase);
      const int r = i & 15;
      const float vi = reinterpret_cast<const float *>(src)[r];
      if (!(vi >= t)) continue;
      const unsigned long long my = ((unsigned long long)fb_f32_ordered(vi) << 32) | (unsigned)(cbase + (r & 3) + 8 * (r >> 2));
      int rank = 0;
      for (int q = 0; q < total * 16 && rank < nsel; ++q) {
        const float4 *s2; int cb2;
        record(q >> 4, s2, cb2);
        const int r2 = q & 15;
        const float vq = reinterpret_cast<const float *>(s2)[r2];
        const unsigned long long kq = ((unsigned long long)fb_f32_ordered(vq) << 32) | (unsigned)(cb2 + (r2 & 3) + 8 * (r2 >> 2));
        rank += kq > my ? 1 : 0;
      }
      if (rank < nsel) {
        const int comp = (int)(unsigned)(my & 0xffffffffull);
        sel[(size_t)row * nsel + rank] = comp;
        if (cnt) atomicAdd(&s_hist[comp], 1);
      }
    }
  }
  for (int s2 = n + l; s2 < nsel; s2 += 16) {   // (NaN features: any valid index)
    sel[(size_t)row * nsel + s2] = s2 % C;
    if (cnt) atomicAdd(&s_hist[s2 % C], 1);
  }
  }
  if (cnt) {
    __syncthreads();
    for (int i = threadIdx.x; i < Cpad; i += 1024) cnt[(size_t)blockIdx.x * Cpad + i] = s_hist[i];
  }
}
void fb_launch_gsel_wide(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap, int n_chunks,
                         int nsel, float *gmax, float *tau, float *gval, unsigned char *gid, int *gcnt, int *flag, int *sel, int *cnt,
                         int Cpad) {
  if (rows_cap <= 0) return;
  const int tiles_a = fb_gsel_w_tiles_a(g, n_chunks, nsel);
  const int NG = 2 * n_chunks * tiles_a, tb = (rows_cap + 15) / 16;
  fb_launch_gsel_w(s, g, feats, n_rows_ptr, rows_cap, n_chunks, 0, gmax, nullptr, nullptr, nullptr, nullptr, tiles_a);
  if (NG <= 64) hipLaunchKernelGGL(k_gsel_tau<4>, dim3(tb), dim3(256), 0, s, gmax, NG, n_rows_ptr, nsel, tau, flag);
  else if (NG <= 128) hipLaunchKernelGGL(k_gsel_tau<8>, dim3(tb), dim3(256), 0, s, gmax, NG, n_rows_ptr, nsel, tau, flag);
  else hipLaunchKernelGGL(k_gsel_tau<16>, dim3(tb), dim3(256), 0, s, gmax, NG, n_rows_ptr, nsel, tau, flag);
  fb_launch_gsel_w(s, g, feats, n_rows_ptr, rows_cap, n_chunks, 1, nullptr, tau, gval, gid, gcnt, tiles_a);
  hipLaunchKernelGGL(k_gsel_final_w, dim3((rows_cap + FB_GSEL_FB - 1) / FB_GSEL_FB), dim3(1024), cnt ? sizeof(int) * (size_t)Cpad : 0, s, gval, gid,
                     gcnt, tau, n_chunks, g.n_tiles / n_chunks, n_rows_ptr, nsel, g.C, sel, flag, cnt, Cpad);
}

void fb_launch_gmm_dump(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                        int rows_cap, int n_chunks, float *ll) {
  if (rows_cap <= 0) return;
  const int tpc = (g.n_tiles + n_chunks - 1) / n_chunks;
  if (g.mode == FB_GMM_MODE_FX2) launch_gmm_fx<true>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, ll, nullptr);
  else launch_gmm_bx<true>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, ll, nullptr);
}

void fb_launch_gmm(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                   int rows_cap, int n_chunks, float *part_m, float *part_s) {
  if (rows_cap <= 0) return;
  const int tpc = (g.n_tiles + n_chunks - 1) / n_chunks;
  if (fb_gmm_use_wide(g)) fb_launch_gmm_wide(s, g, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s);
  else if (g.mode == FB_GMM_MODE_FX2) launch_gmm_fx<false>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s);
  else launch_gmm_bx<false>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s);
}

// raw[b][m] = (1/Tv) * sum_{voiced rows of b} logsumexp_k ll_k   (float64 sum of
// float32 per-frame values, fixed order)
__global__ __launch_bounds__(256) void k_gmm_finalize(FbGmmDev g, const float *__restrict__ part_m,
                                                      const float *__restrict__ part_s, int rows_cap,
                                                      int n_chunks, const int *__restrict__ row_off, int B,
                                                      double *__restrict__ raw) {
  const int b = blockIdx.x, m = blockIdx.y;
  const int r0 = row_off[b], r1 = row_off[b + 1];
  __shared__ double red[256];
  double acc = 0.0;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    float mx = FB_GMM_NEG;
    double ssum = 0.0;
    constexpr int MC = 8;
    if (n_chunks <= MC) {  // every partial of the row requested at once (a loop over a run-time count walks them one
      float pm[MC], ps[MC];  // L2 round trip at a time); same operations in the same order
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const size_t o = ((size_t)min(c, n_chunks - 1) * g.M + m) * rows_cap + r;
        pm[c] = part_m[o];
        ps[c] = part_s[o];
      }
#pragma unroll
      for (int c = 0; c < MC; ++c) if (c < n_chunks) mx = fmaxf(mx, pm[c]);
#pragma unroll
      for (int c = 0; c < MC; ++c) if (c < n_chunks) ssum += (double)ps[c] * exp((double)(pm[c] - mx));
    } else {
      for (int c = 0; c < n_chunks; ++c) mx = fmaxf(mx, part_m[((size_t)c * g.M + m) * rows_cap + r]);
      for (int c = 0; c < n_chunks; ++c) {
        const size_t o = ((size_t)c * g.M + m) * rows_cap + r;
        ssum += (double)part_s[o] * exp((double)(part_m[o] - mx));
      }
    }
    const float ll = (float)((double)mx + log(ssum));
    acc += (double)ll;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int tv = r1 - r0;
    double avg = tv > 0 ? red[0] / (double)tv : __longlong_as_double(0x7ff8000000000000ll);
    if (g.text_scores && tv > 0) avg = fb_round6(avg);
    raw[(size_t)b * g.M + m] = avg;
  }
}

// k_gmm_finalize + k_loss in one launch (GMM systems inside the NES loop): every workgroup finishes one (utterance,
// model) average as k_gmm_finalize does; the workgroup that finishes last (device counter, left at zero) then runs the
// loss / loop-control body on the complete raw matrix.  Same arithmetic and orders as the two kernels.
// 256 threads: with 512 an utterance of up to 512 voiced frames would take one pass instead of two, but the kernel is
// no faster (its time is the last workgroup's chain of small global round trips) and the wider workgroups get in the
// way of the other attacks' kernels: 7.3 k against 8.0 k it/s with three attacks in flight, measured.
#define FB_FIN_THREADS 256
// (the body of one workgroup: utterance b, model m, n_arrive workgroups in all; 256 or 512 threads)
template <bool SMALL>
__device__ __forceinline__ void fb_gmm_finalize_loss_body(const FbGmmDev &g, const float *__restrict__ part_m,
                                                           const float *__restrict__ part_s, int rows_cap,
                                                           int n_chunks, const int *__restrict__ row_off, int B,
                                                           double *__restrict__ raw, int *__restrict__ counter,
                                                           const int *__restrict__ tv, int task, int attack_type,
                                                           const double *__restrict__ z_mean,
                                                           const double *__restrict__ z_std, double threshold,
                                                           double adver_thresh, int target, int true_label,
                                                           const double *__restrict__ dist_part, int n_dist_part,
                                                           double *__restrict__ scores, double *__restrict__ loss,
                                                           FbNesDev *__restrict__ out, FbCtlDev *__restrict__ ctl,
                                                           double *__restrict__ trace, int it, const int b, const int m,
                                                           const int n_arrive, const int pub_seq,
                                                           unsigned long long *__restrict__ xch = nullptr, const bool consumer = false) {
  if (ctl && ctl->stop) return;  // queued behind the stopping iteration
  FN_STAMP(0);
  const int r0 = row_off[b], r1 = row_off[b + 1];
  __shared__ double red[256];
  __shared__ int s_last;
  // the workgroup's threads work out the frame log-likelihoods (four float64 exp and a log each), threads 0 .. 255
  // then add them up in a fixed order: thread t takes frames t, t + 256, ... (the average does not depend on the
  // workgroup size)
  // (SMALL -- samples_per_draw <= 128, every NES batch of the recipe --: the buffers at what such a batch needs, 13 KB
  //  instead of 42: the workgroups are meant to fit beside other attacks' kernels)
  constexpr int FB_LL_LDS = SMALL ? 1024 : 2048;
  constexpr int LV_CAP = SMALL ? 136 : FB_LOSS_LDS, SC_CAP = SMALL ? 768 : FB_SC_LDS;
  __shared__ float s_ll[FB_LL_LDS];
  __shared__ double s_lv[LV_CAP], s_sc[SC_CAP];  // the loss body's (one workgroup of the launch runs it)
  const bool wide = r1 - r0 <= FB_LL_LDS;
  double acc = 0.0;
  for (int r = r0 + threadIdx.x; r < r1; r += (wide ? (int)blockDim.x : 256)) {
    if (!wide && threadIdx.x >= 256) break;

    float mx = FB_GMM_NEG;
    double ssum = 0.0;
    constexpr int MC = 8;
    if (n_chunks <= MC) {  // every partial of the row requested at once (a loop over a run-time count walks them one
      float pm[MC], ps[MC];  // L2 round trip at a time); same operations in the same order
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const size_t o = ((size_t)min(c, n_chunks - 1) * g.M + m) * rows_cap + r;
        pm[c] = part_m[o];
        ps[c] = part_s[o];
      }
#pragma unroll
      for (int c = 0; c < MC; ++c) if (c < n_chunks) mx = fmaxf(mx, pm[c]);
#pragma unroll
      for (int c = 0; c < MC; ++c) if (c < n_chunks) ssum += (double)ps[c] * exp((double)(pm[c] - mx));
    } else {
      for (int c = 0; c < n_chunks; ++c) mx = fmaxf(mx, part_m[((size_t)c * g.M + m) * rows_cap + r]);
      for (int c = 0; c < n_chunks; ++c) {
        const size_t o = ((size_t)c * g.M + m) * rows_cap + r;
        ssum += (double)part_s[o] * exp((double)(part_m[o] - mx));
      }
    }
    const float ll = (float)((double)mx + log(ssum));
    if (wide) s_ll[r - r0] = ll;
    else acc += (double)ll;
  }
  FN_STAMP(1);
  if (wide) {
    __syncthreads();
    if (threadIdx.x < 256)
      for (int r = threadIdx.x; r < r1 - r0; r += 256) acc += (double)s_ll[r];
  }
  if (threadIdx.x < 256) red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  FN_STAMP(2);
  if (xch) {
    // Round 5, the fused launch: no arrival counter.  Every workgroup leaves its score in its exchange slot (an
    // agent-scope store, nothing to wait for) and is done; the LAST-INDEXED workgroup -- every other one was dispatched
    // before it -- polls the B x M slots until none holds the sentinel, puts the sentinels back for the next launch and
    // runs the loss body on the values.  (The counter cost a store round trip, then an atomic round trip, then the
    // queue of 306 atomics on one word: the loss body started ~8.2 us into the launch with the last score formed at 5.3.)
    if (threadIdx.x == 0) {
      const int tvb = r1 - r0;
      double avg = tvb > 0 ? red[0] / (double)tvb : __longlong_as_double(0x7ff8000000000000ll);
      if (g.text_scores && tvb > 0) avg = fb_round6(avg);
      raw[(size_t)b * g.M + m] = avg;
      unsigned long long bits = (unsigned long long)__double_as_longlong(avg);
      if (avg != avg) bits = 0x7ff8000000000000ull;   // (never the sentinel)
      __hip_atomic_store(xch + (size_t)b * g.M + m, bits, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    }
    FN_STAMP(3);
    if (!consumer) return;
    double *s_raw = reinterpret_cast<double *>(s_ll);   // FB_LL_LDS floats: free now (B M <= FB_LL_LDS / 2, checked by the caller)
    __syncthreads();                                     // (every thread is past its reads of s_ll)
    // (touching the control block, tv and the distance partials here, ahead of the body's own loads: no gain, measured)
    for (int i = threadIdx.x; i < B * g.M; i += blockDim.x) {
      unsigned long long v;
      for (;;) {
        v = __hip_atomic_load(xch + i, FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
        if (v != FB_VAD_SENTINEL) break;
        __builtin_amdgcn_s_sleep(1);
      }
      s_raw[i] = __longlong_as_double((long long)v);
      __hip_atomic_store(xch + i, FB_VAD_SENTINEL, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    FN_STAMP(4);
    fb_loss_body<SMALL, false>(s_raw, tv, B, g.M, task, 0, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                               true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it, s_lv, s_sc, pub_seq, LV_CAP, SC_CAP);
    FN_STAMP(5);
    return;
  }
  if (threadIdx.x == 0) {
    const int tvb = r1 - r0;
    double avg = tvb > 0 ? red[0] / (double)tvb : __longlong_as_double(0x7ff8000000000000ll);
    if (g.text_scores && tvb > 0) avg = fb_round6(avg);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(raw + (size_t)b * g.M + m),
                       (unsigned long long)__double_as_longlong(avg), FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    // The score was stored with an agent-scope (write-through) atomic, the arrival counter is an agent-scope atomic and
    // the last arriver reads the scores with agent-scope loads: what has to be ordered is only this thread's store
    // before its own increment -- wait for the store to complete.  (A device-wide release fence here writes back the
    // whole L2 of the XCD, an acquire fence on the other side invalidates it: 306 workgroups x both were most of this
    // kernel's time on the eight-XCD MI355X.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    s_last = (atomicAdd(counter, 1) == n_arrive - 1);
  }
  FN_STAMP(3);
  __syncthreads();
  if (!s_last) return;
  FN_STAMP(4);
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
  // (stamps, tools/profile/fin_instrumented.sh: the body takes ~6 us -- 1.5 until the raw scores are there, ~2.5 forming
  //  the losses, 0.7 barrier, 0.7 one lane's mean, 0.7 decisions, 0.3 publication.  A rehearsal pass without stores ran
  //  first to see whether cold instruction fetch is behind it: the second pass took 5.4 us, so it is not.)
  fb_loss_body<SMALL, true>(raw, tv, B, g.M, task, 0, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                            true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it, s_lv, s_sc, pub_seq, LV_CAP, SC_CAP);
  FN_STAMP(5);
}
#ifdef FB_FIN_STAMP
extern "C" int fb_debug_fin_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_stamps), sizeof(g_fin_stamps)) == hipSuccess ? 0 : -1;
}
#endif
template <bool SMALL>
__global__ __launch_bounds__(FB_FIN_THREADS) void k_gmm_finalize_loss(FbGmmDev g, const float *__restrict__ part_m,
                                                           const float *__restrict__ part_s, int rows_cap,
                                                           int n_chunks, const int *__restrict__ row_off, int B,
                                                           double *__restrict__ raw, int *__restrict__ counter,
                                                           const int *__restrict__ tv, int task, int attack_type,
                                                           const double *__restrict__ z_mean,
                                                           const double *__restrict__ z_std, double threshold,
                                                           double adver_thresh, int target, int true_label,
                                                           const double *__restrict__ dist_part, int n_dist_part,
                                                           double *__restrict__ scores, double *__restrict__ loss,
                                                           FbNesDev *__restrict__ out, FbCtlDev *__restrict__ ctl,
                                                           double *__restrict__ trace, int it) {
  fb_gmm_finalize_loss_body<SMALL>(g, part_m, part_s, rows_cap, n_chunks, row_off, B, raw, counter, tv, task, attack_type, z_mean,
                                   z_std, threshold, adver_thresh, target, true_label, dist_part, n_dist_part, scores, loss, out,
                                   ctl, trace, it, (int)blockIdx.x, (int)blockIdx.y, (int)(gridDim.x * gridDim.y), 0);
}
// ... and the momentum sign step + the next iteration's perturbed batch in the SAME launch (round 5): B x M workgroups
// finalise and -- the last of them -- run the loss body, the N / 256 workgroups behind them are k_update_perturb's
// (fb_update_perturb_body<WAIT>): they stage this iteration's normals and draw the next iteration's while the scores are
// finalised, then wait for the loss body's publication (ctl->pub_seq, agent-scope stores / loads: no device-wide
// fence) and go on.  One launch boundary less in a lone attack's chain, and the Philox + Box-Muller work -- most of
// k_update_perturb -- off its critical path.
// Who waits for whom, and why that cannot deadlock (round-5 advisor finding, settled in round 6): the update workgroups spin
// on the loss body's publication, the consumer -- the last-indexed finalising workgroup -- on the finalisers' slots; the
// finalisers themselves never wait.  Roles follow blockIdx: the finalisers come first, and the hardware dispatches in index
// order, so in practice nobody waits for a workgroup that is not running.  HIP does not promise that order -- but a deadlock
// needs more: the spinners would have to hold EVERY slot of the chip while a finaliser is still undispatched.  A launch has
// at most FB_FUSE_MAX_UPD_WG + 1 = 193 spinners (fb_engine.hip keeps longer audio on k_update_perturb), the fused chain is
// what up to two attacks per GPU run (three with fb_set_fused_chain forced: 579), and the chip has at least 768 slots for
// these 512-thread workgroups (three per compute unit at 42 KB of LDS); every other kernel that may hold slots finishes
// without waiting for this launch.  So a free slot always comes up and every finaliser starts, whatever the order.
// (FB_FIN_TICKET=1 draws the roles from an arrival ticket instead -- finalising roles first by construction; its 494
// returning atomics on one word cost a lone attack 6.8 us per iteration: 0.1319 against 0.1251 ms, tools/profile/r06_fin.sh.)
template <bool SMALL>
__global__ __launch_bounds__(512) void k_gmm_finalize_loss_update(FbGmmDev g, const float *__restrict__ part_m,
                                                           const float *__restrict__ part_s, int rows_cap,
                                                           int n_chunks, const int *__restrict__ row_off, int B,
                                                           double *__restrict__ raw, int *__restrict__ counter,
                                                           const int *__restrict__ tv, int task, int attack_type,
                                                           const double *__restrict__ z_mean,
                                                           const double *__restrict__ z_std, double threshold,
                                                           double adver_thresh, int target, int true_label,
                                                           const double *__restrict__ dist_part, int n_dist_part,
                                                           double *__restrict__ scores, double *__restrict__ loss,
                                                           FbNesDev *__restrict__ out, FbCtlDev *__restrict__ ctl,
                                                           double *__restrict__ trace, int it, int pub_seq, FbUpdArgs u) {
  extern __shared__ double s_dyn_upd[];
  const int n_fin = B * g.M;
  int lin = (int)blockIdx.x;
  if (u.role_ticket) {
    // FB_FIN_TICKET=1: roles by ARRIVAL -- the first B x M arrivals finalise (the last of THEM is the consumer: every slot
    // it polls belongs to an earlier ticket), the later ones update.  Every workgroup draws, stopping launch or not; the
    // last ticket leaves the word at zero for the next launch.
    __shared__ int s_role;
    if (threadIdx.x == 0) {
      const int t = __hip_atomic_fetch_add(u.role_ticket, 1, FB_XCH_RMW, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (int)gridDim.x - 1) __hip_atomic_store(u.role_ticket, 0, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
      s_role = t;
    }
    __syncthreads();
    lin = s_role;
  }
  if (lin < n_fin) {
    fb_gmm_finalize_loss_body<SMALL>(g, part_m, part_s, rows_cap, n_chunks, row_off, B, raw, counter, tv, task, attack_type, z_mean,
                                     z_std, threshold, adver_thresh, target, true_label, dist_part, n_dist_part, scores, loss, out,
                                     ctl, trace, it, lin % B, lin / B, n_fin, pub_seq, n_fin <= (SMALL ? 512 : 1024) ? u.xch : nullptr, lin == n_fin - 1);
  } else {
    fb_update_perturb_body<SMALL, true>(u.loss, u.N, u.half, u.sigma, u.zbuf, u.momentum, u.one_minus_m, u.epsilon, u.audio, u.grad_m,
                                        u.adver, ctl, u.seed, u.next_iter, u.stream, u.q, u.dist_part, u.qscale, lin - n_fin, pub_seq,
                                        s_dyn_upd);
  }
}

void fb_launch_gmm_finalize_loss(hipStream_t s, const FbGmmDev &g, const float *part_m, const float *part_s,
                                 int rows_cap, int n_chunks, const int *row_off, int B, double *raw, int *counter,
                                 const int *tv, int task, int attack_type, const double *z_mean, const double *z_std,
                                 double threshold, double adver_thresh, int target, int true_label,
                                 const double *dist_part, int n_dist_part, double *scores, double *loss, FbNesDev *out,
                                 FbCtlDev *ctl, double *trace, int it, int pub_seq, const FbUpdArgs *upd) {
  if (upd) {   // finalisation + loss + the update / next batch in one launch
    const int blocks = (int)((upd->N + 255) / 256);
    const size_t shm = sizeof(double) * (size_t)(2 * upd->half + 256) + sizeof(float) * 256 * (size_t)(upd->half > 0 ? upd->half : 1);
    {  // (the finalising workgroups' ~34 KB of static LDS + the update's normals: past 64 KB from samples_per_draw = 54 on)
      static std::atomic<unsigned long long> optin{0};
      unsigned long long bit = 0;
      if (fb_device_needs_optin(optin, &bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gmm_finalize_loss_update<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gmm_finalize_loss_update<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        optin.fetch_or(bit, std::memory_order_release);
      }
    }
    if (B - 1 <= 128)
      hipLaunchKernelGGL(k_gmm_finalize_loss_update<true>, dim3(B * g.M + blocks), dim3(512), shm, s, g, part_m, part_s, rows_cap,
                         n_chunks, row_off, B, raw, counter, tv, task, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                         true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it, pub_seq, *upd);
    else
      hipLaunchKernelGGL(k_gmm_finalize_loss_update<false>, dim3(B * g.M + blocks), dim3(512), shm, s, g, part_m, part_s, rows_cap,
                         n_chunks, row_off, B, raw, counter, tv, task, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                         true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it, pub_seq, *upd);
    return;
  }
  if (B - 1 <= 128)
    hipLaunchKernelGGL(k_gmm_finalize_loss<true>, dim3(B, g.M), dim3(FB_FIN_THREADS), 0, s, g, part_m, part_s, rows_cap, n_chunks,
                       row_off, B, raw, counter, tv, task, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                       true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it);
  else
    hipLaunchKernelGGL(k_gmm_finalize_loss<false>, dim3(B, g.M), dim3(FB_FIN_THREADS), 0, s, g, part_m, part_s, rows_cap, n_chunks,
                       row_off, B, raw, counter, tv, task, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                       true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it);
}

void fb_launch_gmm_finalize(hipStream_t s, const FbGmmDev &g, const float *part_m, const float *part_s,
                            int rows_cap, int n_chunks, const int *row_off, int B, double *raw) {
  hipLaunchKernelGGL(k_gmm_finalize, dim3(B, g.M), dim3(256), 0, s, g, part_m, part_s, rows_cap, n_chunks,
                     row_off, B, raw);
}

// ------------------------------------------------------------------------------------------------
// Enrolment (build_spk_models.py:184-216): `gmm-global-acc-stats --update-flags=m` = per-frame posteriors
// of the UBM components (float32 soft-max of the component log-likelihoods, Kaldi's ComponentPosteriors)
// accumulated in float64: occ[k] = sum_t p_tk, F[k][:] = sum_t p_tk x_t, frames in order.
//   k_gmm_lse          wave = frame: max and sum(exp) over the C log-likelihoods of the dump matrix
//   k_gmm_post_stats   workgroup = 64-component slab, thread (c, dg) owns component c and dims dg, dg+4, ...;
//                      frames are staged 64 at a time (posterior tile + feature tile in LDS)
__global__ __launch_bounds__(256) void k_gmm_lse(int C, int ld, const float *__restrict__ ll,
                                                 const int *__restrict__ n_rows_ptr, float *__restrict__ mx,
                                                 float *__restrict__ inv_sum) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= *n_rows_ptr) return;
  const float *lr = ll + (size_t)row * ld;
  float m = -FLT_MAX;
  for (int i = lane; i < C; i += 64) m = fmaxf(m, lr[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float ssum = 0.0f;
  for (int i = lane; i < C; i += 64) ssum += expf(lr[i] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
  if (lane == 0) { mx[row] = m; inv_sum[row] = 1.0f / ssum; }
}
#define FB_PS_DMAX4 20  // D <= 80
__global__ __launch_bounds__(256) void k_gmm_post_stats(int C, int ld, int D, const float *__restrict__ ll,
                                                        const float *__restrict__ feats,
                                                        const int *__restrict__ n_rows_ptr,
                                                        const float *__restrict__ mx,
                                                        const float *__restrict__ inv_sum,
                                                        double *__restrict__ occ, double *__restrict__ F) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  float *Pd = smf;           // [64 rows][64 comps]
  float *X = smf + 64 * 64;  // [64 rows][D]
  const int n_rows = *n_rows_ptr, k0 = blockIdx.x * 64;
  const int c = threadIdx.x & 63, dg = threadIdx.x >> 6;
  double acc[FB_PS_DMAX4];
#pragma unroll
  for (int i = 0; i < FB_PS_DMAX4; ++i) acc[i] = 0.0;
  double gam = 0.0;
  for (int rb = 0; rb < n_rows; rb += 64) {
    const int nr = min(64, n_rows - rb);
    __syncthreads();
    for (int i = threadIdx.x; i < nr * 64; i += 256) {
      const int rl = i >> 6, cc = i & 63;
      const int k = k0 + cc;
      // Kaldi: exp(ll - max) scaled by 1/sum, all in float32
      Pd[i] = k < C ? expf(ll[(size_t)(rb + rl) * ld + k] - mx[rb + rl]) * inv_sum[rb + rl] : 0.0f;
    }
    for (int i = threadIdx.x; i < nr * D; i += 256) X[i] = feats[(size_t)rb * D + i];
    __syncthreads();
    for (int rl = 0; rl < nr; ++rl) {
      const double wv = (double)Pd[rl * 64 + c];
      gam = __dadd_rn(gam, wv);
      const float *fr = X + rl * D;
#pragma unroll
      for (int i = 0; i < FB_PS_DMAX4; ++i) {
        const int d = dg + 4 * i;
        if (d < D) acc[i] = __dadd_rn(acc[i], __dmul_rn(wv, (double)fr[d]));
      }
    }
  }
  const int k = k0 + c;
  if (k < C) {
    if (dg == 0) occ[k] = gam;
#pragma unroll
    for (int i = 0; i < FB_PS_DMAX4; ++i) {
      const int d = dg + 4 * i;
      if (d < D) F[(size_t)k * D + d] = acc[i];
    }
  }
}
void fb_launch_gmm_post_stats(hipStream_t s, int C, int ld, int D, const float *ll, const float *feats,
                              const int *n_rows_ptr, int rows_cap, float *mx, float *inv_sum, double *occ,
                              double *F) {
  if (rows_cap <= 0) return;
  hipLaunchKernelGGL(k_gmm_lse, dim3((rows_cap + 3) / 4), dim3(256), 0, s, C, ld, ll, n_rows_ptr, mx, inv_sum);
  hipLaunchKernelGGL(k_gmm_post_stats, dim3((C + 63) / 64), dim3(256), sizeof(float) * (64 * 64 + 64 * (size_t)D), s, C,
                     ld, D, ll, feats, n_rows_ptr, mx, inv_sum, occ, F);
}
