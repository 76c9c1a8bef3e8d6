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

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FB_GMM_NEG (-3.0e38f)

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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fb_pow2f(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }

__device__ __forceinline__ void fb_split2_frag(const float (&v)[8], u32x4 &f1, u32x4 &f2) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16x2 a, b;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float x = v[2 * i + u];
      const _Float16 x1 = (_Float16)x;                                   // round to nearest even
      const float r = __fsub_rn(x, (float)x1);  // exact; possibly a subnormal f16, which the matrix pipe keeps
      a[u] = x1;
      b[u] = (_Float16)r;
    }
    f1[i] = __builtin_bit_cast(unsigned, a);
    f2[i] = __builtin_bit_cast(unsigned, b);
  }
}

#define FB_FX_MFMA(A, B, ACC) \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), ACC, 0, 0, 0)

// online logsumexp of k_gmm_fx2: fold 16 values into the state (m, s) at (stm, sts).
// The state lives in the log2 domain so that one value costs one (packed) fma, one v_exp_f32 and one (packed) add:
//   m = running maximum (a value, exact),  r = fl(m * L),  s = sum 2^(v * L - r),   L = fl(log2 e)
//   => logsumexp = ln2 * (r + log2 s); fb_lse_to_natural() converts to the (m, sum exp(v - m)) convention the
//   chunk merge / k_gmm_finalize use.  The reference point r only has to be the same for every term of s.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define FB_LOG2E_F 1.44269502162933349609375f  // fl(log2 e)
__device__ __forceinline__ void fb_lse_update16(const f32x16 &pv, float *__restrict__ stm, float *__restrict__ sts,
                                                float ls = FB_LOG2E_F) {  // ls = fl(log2 e) * 2^-kacc for scaled values
  float tm = FB_GMM_NEG;
#pragma unroll
  for (int r = 0; r < 16; ++r) tm = fmaxf(tm, pv[r]);
  const float m_old = *stm, s_old = *sts;
  const float m_new = fmaxf(m_old, tm);
  const float r_new = __fmul_rn(m_new, ls), r_old = __fmul_rn(m_old, ls);  // r_old = -inf at the start
  const f32x2 l2 = {ls, ls}, nr2 = {-r_new, -r_new};
  f32x2 acc = {s_old * __builtin_amdgcn_exp2f(r_old - r_new), 0.0f};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const f32x2 v2 = {pv[2 * r], pv[2 * r + 1]};
    const f32x2 t = __builtin_elementwise_fma(v2, l2, nr2);
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    acc += e;
  }
  *stm = m_new;
  *sts = acc[0] + acc[1];
}
// s (log2-domain state, see above) -> sum exp(v - m):  s * 2^(r - L m), evaluated in float64 (|r - L m| < 1e-4)
__device__ __forceinline__ float fb_lse_to_natural(float m, float s, float ls = FB_LOG2E_F) {
  const double d = (double)__fmul_rn(m, ls) - (double)ls * (double)m;
  return (float)((double)s * (1.0 + d * 0.6931471805599453 * (1.0 + d * 0.34657359027997264)));
}

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

template <int NK, bool DUMP>
__global__ __launch_bounds__(256, FB_FX_OCC) void k_gmm_fx2(FbGmmDev g, const float *__restrict__ feats,
                                                    const int *__restrict__ n_rows_ptr, int tiles_per_chunk,
                                                    int rows_cap, float *__restrict__ part_m,
                                                    float *__restrict__ part_s, int xcd_map) {
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
  int sh = 0;
  {
    const bool ok = row < n_rows;
    const float *fr = feats + (size_t)(ok ? row : 0) * g.D;
    const float qs = fb_pow2f(g.kx2), xs = fb_pow2f(g.kx);  // exact power-of-two operand scalings
    float vv[NK][8], qq[NK][8];
    float amax = xs;
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      const int d0 = 16 * c + 8 * h;
      float *v = vv[c], *q = qq[c];
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
      for (int i = 0; i < 8; ++i) q[i] = __fmul_rn(__fmul_rn(v[i], v[i]), qs);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (d0 + i == g.D) ? xs : __fmul_rn(v[i], xs);
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fmaxf(fabsf(v[i]), q[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (amax >= 32768.0f) {  // wave-uniform; finite features only (the front-end produces nothing else)
      const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;  // amax in [2^ex, 2^(ex+1))
      sh = min(ex - 14, 100);
    }
    const float down = fb_pow2f(-sh);
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      if (sh) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { vv[c][i] = __fmul_rn(vv[c][i], down); qq[c][i] = __fmul_rn(qq[c][i], down); }
      }
      fb_split2_frag(vv[c], bx1[c], bx2[c]);
      fb_split2_frag(qq[c], bq1[c], bq2[c]);
    }
  }
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
        if (row < n_rows) {
          const int tile = tile0 + it / g.n_items;
          float *dst = part_m + (size_t)row * (g.n_tiles * 32) + tile * 32 + 4 * h;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            *reinterpret_cast<float4 *>(dst + 8 * rr) = make_float4(pv[4 * rr] * unscale, pv[4 * rr + 1] * unscale,
                                                                    pv[4 * rr + 2] * unscale, pv[4 * rr + 3] * unscale);
        }
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


// ---------------------------------------------------------------------------------------------
// k_gmm_fx2w: the scoring form of k_gmm_fx2 for ONE variance group (mean-only MAP adaptation: the UBM and all its
// speaker models; items per tile = Q, m_0 .. m_{M-1}) -- one wave per SIMD, 64 frames per wave, software-pipelined,
// the whole tile as straight-line code.
//
// What the design rests on (tools/probes/coissue_probe.hip, valu_cost_probe.hip; one wave per SIMD):
//   * MFMAs on ONE accumulator issue only as fast as they execute (the wave sits at the next dependent MFMA), so in
//     k_gmm_fx2 an item's update starts when its 15 MFMAs are done.  MFMAs that alternate between two INDEPENDENT
//     accumulators are queued by the matrix pipe and the wave goes on issuing behind them.
//   * The wave issues in order: a vector instruction overlaps the matrix pipe only if it stands behind an MFMA in the
//     instruction stream.  About five plain VALU instructions per MFMA are free (16.5 ns per MFMA with 0 .. 4
//     v_fma_f32 behind it, 18.5 with 6, 23 with 8); 2 v_fma + 2 v_exp + 2 v_add cost 20 ns.
//   * A PACKED f32 instruction behind an MFMA stalls the wave: MFMA + one v_pk_fma_f32 = 20.6 ns, the packed form of
//     the update's gap 30.7 ns against 20.3 ns unpacked.  The update is written with single instructions.
// The two independent chains of an item are the two 32-frame halves of the wave's 64 frames: both use the SAME
// parameter fragments (half the LDS reads per MFMA) and every item, the quadratic one included, is a pair.  That needs
// 160 registers of frame operands + 96 of accumulators (hq, in flight, being consumed; two halves each) + 32 for
// the values being consumed + 24 of parameter fragments: the 512-register budget of one wave per SIMD -- 4 waves
// (256 frames) per workgroup, one workgroup per CU, component chunks chosen so that a launch is one round of <= 256
// workgroups.  An empty asm pins the accumulators to AGPRs (left alone hipcc parks the FRAME operands there and
// copies them back in front of every MFMA: 1190 v_accvgpr moves).
// With one wave per SIMD nothing hides instruction fetch after a branch (a loop over items with the item kind,
// pending update and padding decided by branches ran at ~1500 cycles per item with an EMPTY body), hence the
// specialisation: M is a template parameter, accumulator sets have static roles (model m writes set m & 1 while set
// (m - 1) & 1 is updated in the gaps between its MFMAs, fb_fxw_step; the quadratic item carries the update of the
// previous tile's last model), and a tile is one basic block.  Models with C % 32 != 0, several variance groups or
// other M run on k_gmm_fx2.
// Parameter items arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into two LDS
// slots, a group of items each, requested a whole group of steps ahead of the barrier that publishes them
// (fb_fxw_fetch and the loop below).
__device__ __forceinline__ void fb_glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// N (1 .. 4) consecutive 1 KB pieces with one M0 set-up: the instruction offset moves the global source AND the LDS
// destination (tools/probes/glds_offset_probe.hip)
template <int N>
__device__ __forceinline__ void fb_glds16_run(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
#define FB_GLDS_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
#define FB_GLDS_AT(o) "global_load_lds_dwordx4 %1, off offset:" #o "\n\t"
#define FB_GLDS_TAIL "s_mov_b32 m0, %0"
  if constexpr (N == 1)
    asm volatile(FB_GLDS_HEAD FB_GLDS_TAIL : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else if constexpr (N == 2)
    asm volatile(FB_GLDS_HEAD FB_GLDS_AT(1024) FB_GLDS_TAIL : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else if constexpr (N == 3)
    asm volatile(FB_GLDS_HEAD FB_GLDS_AT(1024) FB_GLDS_AT(2048) FB_GLDS_TAIL : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile(FB_GLDS_HEAD FB_GLDS_AT(1024) FB_GLDS_AT(2048) FB_GLDS_AT(3072) FB_GLDS_TAIL
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#undef FB_GLDS_HEAD
#undef FB_GLDS_AT
#undef FB_GLDS_TAIL
}
// A group of NITEMS consecutive parameter items (NPIECE KB each, contiguous in the image buffer) -> LDS: wave wv brings
// the pieces [wv PW, (wv + 1) PW), PW = ceil(total / 4) -- up to 3 pieces past the group's end when the total is not
// a multiple of four (the slots are padded for them, the image buffer is allocated 4 KB longer).
template <int NITEMS, int NPIECE>
__device__ __forceinline__ void fb_fxw_fetch(const u32x4 *__restrict__ group_lane, unsigned lds_dst, int wv) {
  constexpr int TOT = NITEMS * NPIECE, PW = (TOT + 3) / 4;
  const u32x4 *src = group_lane + (size_t)wv * (PW * 64);
  const unsigned dst = lds_dst + (unsigned)wv * (PW * 1024);
#pragma unroll
  for (int u = 0; u < PW; u += 4) {
    const int n = PW - u;
    if (n >= 4) fb_glds16_run<4>(src + u * 64, dst + u * 1024);
    else if (n == 3) fb_glds16_run<3>(src + u * 64, dst + u * 1024);
    else if (n == 2) fb_glds16_run<2>(src + u * 64, dst + u * 1024);
    else fb_glds16_run<1>(src + u * 64, dst + u * 1024);
  }
}

// Single vector instructions, pinned where they are written (volatile): the update slices below must stay in their
// MFMA gaps, and they must NOT be packed -- a v_pk_fma_f32 / v_pk_add_f32 behind an MFMA costs the wave ~4 ns where
// four plain v_fma_f32 are free (tools/probes/valu_cost_probe.hip).
__device__ __forceinline__ float fb_v_fma(float a, float b, float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fb_v_exp(float a) {  // 2^a; its consumer stands at least one gap later (no trans hazard)
  float d;
  asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a));
  return d;
}
__device__ __forceinline__ float fb_v_add(float a, float b) {
  float d;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float fb_v_mul(float a, float b) {
  float d;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float fb_v_max3(float a, float b, float c) {
  float d;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// One item step with the logsumexp update of ANOTHER accumulator set threaded between the MFMAs, by construction.
// One wave per SIMD issues in order; about five plain vector instructions behind an MFMA are free, what exceeds them
// adds to the step.  Left alone hipcc lumps the ~160 vector instructions of an update behind the 30 MFMAs (kernel time =
// MFMA time + update time, measured), so the step is cut into one scheduling region per MFMA
// (__builtin_amdgcn_sched_barrier(0)) and each region gets its slice of the update of the pending set (p0, p1 = the two
// halves' 16 values each, pm / ps their LDS state [2 halves][256]):
//   gap  0       the state is requested from LDS (the pending values are still in the matrix pipe)
//   gaps 2..9    four values are copied from the accumulation registers into vector registers, where they stay for
//                the second pass (two reads per value would make the update the longer pipe); running maximum
//   gaps 10, 11  new reference r = fl(m L), the old sums rescaled
//   gaps 12..27  one value of each half: fma, exponential, and the ADD of the previous gap's exponentials (so that
//                no exponential is consumed right behind itself); even and odd values are summed apart and joined at
//                the end, which is fb_lse_update16's order
//   gaps 28, 29  last adds, state written back
// The parameter fragments are streamed with the K chunks -- chunk c + 1 is read from LDS while the six MFMAs of chunk
// c run (two alternating register sets for the chunks 1 .. NK-1; chunk 0 has its own, z1 / z2, refilled with the NEXT
// item's chunk 0 when pf0) -- 24 registers instead of two whole items' 80: that is what leaves room for the 32
// pending values.
// The LDS-DMA pieces [dq0, dq0 + dn) of this wave's share of the next parameter group go out one per chunk.
// UPD = false: no pending set (the first items of the first tile).
template <int NK, bool UPD>
__device__ __forceinline__ void fb_fxw_step(const u32x4 *__restrict__ cur4, const u32x4 *__restrict__ nxt4, const bool pf0,
                                            int lane, u32x4 &z1, u32x4 &z2, const u32x4 (&b1)[2][NK],
                                            const u32x4 (&b2)[2][NK], const f32x16 &init0, const f32x16 &init1,
                                            f32x16 &out0, f32x16 &out1, const f32x16 &p0, const f32x16 &p1,
                                            float *__restrict__ pm, float *__restrict__ ps, float ls,
                                            const u32x4 *__restrict__ dsrc, unsigned ddst, const int dq0, const int dn) {
  constexpr int NG = 6 * NK;
  static_assert(NG >= 30, "slice layout");
  f32x16 x0 = init0, x1 = init1;
  u32x4 s1[2], s2[2];  // fragment sets of the chunks 1 .. NK-1: chunk c uses set c & 1
  float v0[16], v1[16];
  float t0 = FB_GMM_NEG, t1 = FB_GMM_NEG, mo0 = 0.f, mo1 = 0.f, so0 = 0.f, so1 = 0.f, mn0 = 0.f, mn1 = 0.f;
  float nr0 = 0.f, nr1 = 0.f, d0 = 0.f, d1 = 0.f;
  float se0 = 0.f, se1 = 0.f, sd0 = 0.f, sd1 = 0.f;  // sums of the even / odd values, halves 0 / 1
  float e0 = 0.f, e1 = 0.f;                          // the previous gap's exponentials
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = g / 6, k = g % 6;
    if (k == 0) {
      if (c + 1 < NK) {
        s1[(c + 1) & 1] = cur4[(0 * NK + c + 1) * 64 + lane];
        s2[(c + 1) & 1] = cur4[(1 * NK + c + 1) * 64 + lane];
      }
    }
    const u32x4 &a1 = c == 0 ? z1 : s1[c & 1], &a2 = c == 0 ? z2 : s2[c & 1];
    if (k == 0) FB_FX_MFMA(a2, b1[0][c], x0);
    else if (k == 1) FB_FX_MFMA(a2, b1[1][c], x1);
    else if (k == 2) FB_FX_MFMA(a1, b2[0][c], x0);
    else if (k == 3) FB_FX_MFMA(a1, b2[1][c], x1);
    else if (k == 4) FB_FX_MFMA(a1, b1[0][c], x0);
    else FB_FX_MFMA(a1, b1[1][c], x1);
    if (pf0 && g == 7) {  // chunk 0's registers are free: the next item's chunk 0 (pf0 is a constant after unrolling)
      z1 = nxt4[(0 * NK + 0) * 64 + lane];
      z2 = nxt4[(1 * NK + 0) * 64 + lane];
    }
    if (k == 3 && c < dn) fb_glds16(dsrc + (dq0 + c) * 64, ddst + (unsigned)(dq0 + c) * 1024u);
    if constexpr (UPD) {
      if (g == 0) { mo0 = pm[0]; mo1 = pm[256]; so0 = ps[0]; so1 = ps[256]; }
      if (g >= 2 && g < 10) {
        const int r = 2 * (g - 2);
        v0[r] = p0[r]; v0[r + 1] = p0[r + 1]; v1[r] = p1[r]; v1[r + 1] = p1[r + 1];
        // the copies are the compiler's (it knows the matrix pipe's hazards); the empty asm keeps them HERE and in
        // vector registers
        asm volatile("" : "+v"(v0[r]), "+v"(v0[r + 1]), "+v"(v1[r]), "+v"(v1[r + 1]));
        t0 = fb_v_max3(t0, v0[r], v0[r + 1]);
        t1 = fb_v_max3(t1, v1[r], v1[r + 1]);
      } else if (g == 10) {
        mn0 = fb_v_max3(mo0, t0, t0); mn1 = fb_v_max3(mo1, t1, t1);
        const float rn0 = fb_v_mul(mn0, ls), rn1 = fb_v_mul(mn1, ls);
        nr0 = -rn0; nr1 = -rn1;
        d0 = fb_v_fma(mo0, ls, nr0); d1 = fb_v_fma(mo1, ls, nr1);  // r_old - r_new; r_old = -inf at the start
      } else if (g == 11) {
        e0 = fb_v_exp(d0); e1 = fb_v_exp(d1);
      } else if (g >= 12 && g < 28) {
        const int r = g - 12;
        const float u0 = fb_v_fma(v0[r], ls, nr0), u1 = fb_v_fma(v1[r], ls, nr1);
        const float f0 = fb_v_exp(u0), f1 = fb_v_exp(u1);
        if (r == 0) { se0 = fb_v_mul(so0, e0); se1 = fb_v_mul(so1, e1); }      // s_old * 2^(r_old - r_new)
        else if (r == 1) { se0 = fb_v_add(se0, e0); se1 = fb_v_add(se1, e1); }  // + value 0
        else if (r == 2) { sd0 = e0; sd1 = e1; }                                 // value 1 starts the odd sums
        else if (r & 1) { se0 = fb_v_add(se0, e0); se1 = fb_v_add(se1, e1); }  // value r - 1 is even
        else { sd0 = fb_v_add(sd0, e0); sd1 = fb_v_add(sd1, e1); }
        e0 = f0; e1 = f1;
      } else if (g == 28) {
        sd0 = fb_v_add(sd0, e0); sd1 = fb_v_add(sd1, e1);                       // value 15
      } else if (g == 29) {
        pm[0] = mn0; pm[256] = mn1;
        ps[0] = fb_v_add(se0, sd0); ps[256] = fb_v_add(se1, sd1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("" : "+a"(x0), "+a"(x1));
  out0 = x0;
  out1 = x1;
}

template <int NK, int M>
__global__ __launch_bounds__(256, 1) void k_gmm_fx2w(FbGmmDev g, const float *__restrict__ feats,
                                                     const int *__restrict__ n_rows_ptr, int tiles_per_chunk,
                                                     int rows_cap, float *__restrict__ part_m,
                                                     float *__restrict__ part_s, int xcd_map) {
  if (g.stop && *g.stop) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG4 = 2 * NK * 64;  // 16-byte units per item
  constexpr int NI = M + 1;          // items per tile: Q, m_0 .. m_{M-1}
  constexpr int GA = (NI + 1) / 2, GB = NI - GA;  // a tile's items live in two LDS slots: A = items 0 .. GA-1, B = the rest
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
  const int strip0 = strip_i * 256;
  if (strip0 >= n_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  constexpr int PAD4 = 192;                    // 3 KB behind each slot: fb_fxw_fetch rounds a group up to whole waves
  constexpr int SLOTB4 = GA * IMG4 + PAD4;     // slot A = items 0 .. GA-1 at 0, slot B = the rest
  const u32x4 *slot0 = reinterpret_cast<const u32x4 *>(lds);
  float *st_m = lds + (NI * IMG4 + 2 * PAD4) * 4;  // [M][2 halves][256]
  float *st_s = st_m + M * 512;                // [M][2 halves][256]

  // ---- frame fragments of the two 32-frame halves (layout and range guard as in k_gmm_fx2; the power-of-two shift
  //      is uniform over the wave's 64 frames)
  u32x4 bx1[2][NK], bx2[2][NK], bq1[2][NK], bq2[2][NK];
  int sh = 0;
  int rows[2];
  {
    const float qs = fb_pow2f(g.kx2), xs = fb_pow2f(g.kx);
    float vv[2][NK][8], qq[2][NK][8];
    float amax = xs;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      rows[hf] = strip0 + w * 64 + hf * 32 + j;
      const bool ok = rows[hf] < n_rows;
      const float *fr = feats + (size_t)(ok ? rows[hf] : 0) * g.D;
#pragma unroll
      for (int c = 0; c < NK; ++c) {
        const int d0 = 16 * c + 8 * h;
        float *v = vv[hf][c], *q = qq[hf][c];
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
        for (int i = 0; i < 8; ++i) q[i] = __fmul_rn(__fmul_rn(v[i], v[i]), qs);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (d0 + i == g.D) ? xs : __fmul_rn(v[i], xs);
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fmaxf(fabsf(v[i]), q[i]));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (amax >= 32768.0f) {
      const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
      sh = min(ex - 14, 100);
    }
    const float down = fb_pow2f(-sh);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int c = 0; c < NK; ++c) {
        if (sh) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { vv[hf][c][i] = __fmul_rn(vv[hf][c][i], down); qq[hf][c][i] = __fmul_rn(qq[hf][c][i], down); }
        }
        fb_split2_frag(vv[hf][c], bx1[hf][c], bx2[hf][c]);
        fb_split2_frag(qq[hf][c], bq1[hf][c], bq2[hf][c]);
      }
  }
#pragma unroll
  for (int m = 0; m < 2 * M; ++m) { st_m[m * 256 + tid] = FB_GMM_NEG; st_s[m * 256 + tid] = 0.0f; }

  const int tile0 = chunk_i * tiles_per_chunk;
  const int tile1 = min(g.n_tiles, tile0 + tiles_per_chunk);
  const int n_t = tile1 - tile0, total_items = n_t * NI;
  const u32x4 *gimg = g.images_fx + (size_t)tile0 * NI * IMG4;
  const float unscale = fb_pow2f(sh - g.kacc), ls = __fmul_rn(FB_LOG2E_F, unscale);  // exact: a power of two

  // ---- parameter stream.  A workgroup barrier per item costs ~400 cycles at one wave per SIMD (the probe's mode 12
  // against 11), so the barrier is taken twice per TILE: slot A holds items 0 .. GA-1, slot B items GA .. NI-1.
  // While group A of tile t runs the LDS-DMA fills slot B with group B of the same tile, while group B runs it fills
  // slot A with group A of tile t + 1: a group is one contiguous piece of the image buffer, wave w brings the 1 KB
  // pieces [w PW, (w + 1) PW) of it, five per step (one behind the fourth MFMA of each K chunk) in the group's first
  // two steps -- at least a step ahead of the vmcnt(0) + barrier that publishes the slot (measured with s_memtime
  // stamps: the wait is 16 cycles, the barrier ~90, of ~5 000 per group).
  constexpr int NPIECE = IMG4 / 64;
  const int wv = __builtin_amdgcn_readfirstlane(w);
  const unsigned ring_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)lds;
  auto item4 = [&](int jj) { return jj < GA ? jj * IMG4 : SLOTB4 + (jj - GA) * IMG4; };
  auto update = [&](const f32x16 &v0, const f32x16 &v1, int model) {
    fb_lse_update16(v0, st_m + (2 * model) * 256 + tid, st_s + (2 * model) * 256 + tid, ls);
    fb_lse_update16(v1, st_m + (2 * model + 1) * 256 + tid, st_s + (2 * model + 1) * 256 + tid, ls);
  };
  auto publish = [&]() {  // everything this wave asked for has landed; the barrier publishes all four waves' pieces
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  f32x16 hq[2], acc[2][2], zero;  // acc[set][half]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    hq[0][r] = 0.f; hq[1][r] = 0.f; zero[r] = 0.f;
    acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f;
  }
  {
    // The quadratic step of the FIRST tile has no finished model to carry, and a branch for it costs more than a bogus
    // update (nothing hides an instruction fetch at one wave per SIMD): it "updates" the last model with 16 sentinel
    // values -2^60 / unscale, whose scaled form is exactly -2^60 fl(log2 e) -- the state becomes (that maximum, 16),
    // and the first real update rescales those 16 by 2^(-1.6e18) = 0.
    const float sentinel = -fb_pow2f(60 - sh + g.kacc);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[(M - 1) & 1][0][r] = sentinel; acc[(M - 1) & 1][1][r] = sentinel; }
  }
  fb_fxw_fetch<GA, NPIECE>(gimg + lane, ring_lds, wv);  // group A of the first tile
  publish();
  u32x4 z1, z2;  // chunk 0 of the item in front (fb_fxw_step)
  constexpr int PW_A = (GA * NPIECE + 3) / 4, PW_B = (GB * NPIECE + 3) / 4;  // LDS-DMA pieces per wave for a group
  for (int t = 0; t < n_t; ++t) {
    const int it0 = t * NI;
    // this wave's share of the groups requested during this tile: group B of this tile (while A runs), group A of the
    // next one (while B runs); past the chunk's end the last group is read again and never used
    const u32x4 *srcB = gimg + (size_t)min(it0 + GA, total_items - GB) * IMG4 + (size_t)wv * (PW_B * 64) + lane;
    const u32x4 *srcA = gimg + (size_t)min(it0 + NI, total_items - GA) * IMG4 + (size_t)wv * (PW_A * 64) + lane;
    const unsigned dstB = ring_lds + SLOTB4 * 16 + (unsigned)wv * (PW_B * 1024), dstA = ring_lds + (unsigned)wv * (PW_A * 1024);
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {   // compile-time item index within the tile: 0 = Q, 1 + m = model m
      const bool first_of_group = (jj == 0 || jj == GA);
      const u32x4 *cur4 = slot0 + item4(jj);
      if (first_of_group) {  // its image was published by the barrier just passed: chunk 0 is not prefetched
        z1 = cur4[(0 * NK + 0) * 64 + lane];
        z2 = cur4[(1 * NK + 0) * 64 + lane];
      }
      // the pieces of the other slot's next content: five per step from the group's first step on
      const int gs = jj < GA ? jj : jj - GA;                      // step within the group
      const int pw = jj < GA ? PW_B : PW_A, dq0 = 5 * gs;
      int dn = dq0 < pw ? (pw - dq0 < 5 ? pw - dq0 : 5) : 0;
      const u32x4 *dsrc = jj < GA ? srcB : srcA;
      const unsigned ddst = jj < GA ? dstB : dstA;
      if (first_of_group) {  // behind the chunk-0 reads just issued: their latency is there anyway
#pragma unroll
        for (int q = 0; q < dn; ++q) fb_glds16(dsrc + q * 64, ddst + (unsigned)q * 1024u);
        dn = 0;
      }
      // the MFMAs of this item with the update of the pending accumulator set threaded between them (fb_fxw_step):
      // the quadratic item carries the previous tile's last model, model m >= 1 carries model m - 1
      const bool pf0 = (jj + 1 < NI && jj + 1 != GA);
      const u32x4 *nxt4 = slot0 + item4(jj + 1 < NI ? jj + 1 : 0);
      if (jj == 0) {
        float *pm = st_m + (2 * (M - 1)) * 256 + tid, *ps = st_s + (2 * (M - 1)) * 256 + tid;
        fb_fxw_step<NK, true>(cur4, nxt4, pf0, lane, z1, z2, bq1, bq2, zero, zero, hq[0], hq[1], acc[(M - 1) & 1][0], acc[(M - 1) & 1][1], pm, ps, ls, dsrc, ddst, dq0, dn);
      } else if (jj == 1) {
        fb_fxw_step<NK, false>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[0][0], acc[0][1], zero, zero, st_m, st_s, ls, dsrc, ddst, dq0, dn);
      } else {
        float *pm = st_m + (2 * (jj - 2)) * 256 + tid, *ps = st_s + (2 * (jj - 2)) * 256 + tid;
        fb_fxw_step<NK, true>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[(jj - 1) & 1][0], acc[(jj - 1) & 1][1],
                              acc[(jj - 2) & 1][0], acc[(jj - 2) & 1][1], pm, ps, ls, dsrc, ddst, dq0, dn);
      }
      if (jj == GA - 1 || jj == NI - 1) publish();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  update(acc[(M - 1) & 1][0], acc[(M - 1) & 1][1], M - 1);

#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const float ms = st_m[(2 * m + hf) * 256 + tid];  // maximum of ll * 2^(kacc - sh)
      const float mm = ms * unscale, ss = fb_lse_to_natural(ms, st_s[(2 * m + hf) * 256 + tid], ls);
      const float m2 = __shfl_xor(mm, 32, 64), s2 = __shfl_xor(ss, 32, 64);
      const float mx = fmaxf(mm, m2);
      const float sx = ss * __expf(mm - mx) + s2 * __expf(m2 - mx);
      if (h == 0 && rows[hf] < n_rows) {
        const size_t o = ((size_t)chunk_i * M + m) * rows_cap + rows[hf];
        part_m[o] = mx;
        part_s[o] = sx;
      }
    }
}

template <int NK, int M>
static void launch_gmm_fxw_t(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                             int rows_cap, int n_chunks, int tpc, float *part_m, float *part_s) {
  const int strips = (rows_cap + 255) / 256;
  dim3 grid((unsigned)strips, (unsigned)n_chunks);
  int xcd_map = 0;
  static const bool no_xcd_map = getenv("FB_GMM_NO_XCD_MAP") != nullptr;
  if ((n_chunks == 1 || n_chunks == 2 || n_chunks == 4 || n_chunks == 8) && !no_xcd_map) {
    const int per = 8 / n_chunks;
    grid = dim3((unsigned)(8 * ((strips + per - 1) / per)), 1);
    xcd_map = n_chunks;
  }
  const size_t ldsb = ((size_t)(M + 1) * 2 * NK * 64 + 2 * 192) * 16 + (size_t)2 * M * 512 * sizeof(float);  // one tile (two padded slots) + the state
  static std::atomic<unsigned long long> optin{0};
  unsigned long long bit = 0;
  if (ldsb > 64 * 1024 && fb_device_needs_optin(optin, &bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gmm_fx2w<NK, M>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) == hipSuccess)
      optin.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_gmm_fx2w<NK, M>), grid, dim3(256), ldsb, s, g, feats, n_rows_ptr, tpc, rows_cap, part_m, part_s,
                     xcd_map);
}
// k_gmm_fx2w is instantiated for the shapes the reference's systems have with the recipe's 72-dimensional features
// (NKF = 5): one variance group, every component tile full, 2 <= M <= FB_FXW_MAX_M models (SV: UBM + 1; OSI: UBM +
// speakers; CSI: the speakers).  Everything else runs on k_gmm_fx2.
#define FB_FXW_MAX_M 6
bool fb_gmm_use_wide(const FbGmmDev &g) {
  const bool off = getenv("FB_GMM_NARROW") != nullptr;  // read per call: the tests switch it inside one process
  return g.mode == FB_GMM_MODE_FX2 && !off && g.NKF == 5 && g.n_items == g.M + 1 && (g.C & 31) == 0 && g.M >= 2 &&
         g.M <= FB_FXW_MAX_M && g.item_model_host_q_first;
}
static void launch_gmm_fxw(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap,
                           int n_chunks, int tpc, float *part_m, float *part_s) {
  switch (g.M) {
    case 2: launch_gmm_fxw_t<5, 2>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 3: launch_gmm_fxw_t<5, 3>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 4: launch_gmm_fxw_t<5, 4>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 5: launch_gmm_fxw_t<5, 5>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 6: launch_gmm_fxw_t<5, 6>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    default: break;  // fb_gmm_use_wide() admits only the cases above
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
  const size_t ldsb = (size_t)2 * 2 * NK * 64 * 16 + (size_t)2 * g.M * 256 * sizeof(float);
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
  if (fb_gmm_use_wide(g)) launch_gmm_fxw(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s);
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
  if (ctl && ctl->stop) return;  // queued behind the stopping iteration
  const int b = blockIdx.x, m = blockIdx.y;
  const int r0 = row_off[b], r1 = row_off[b + 1];
  __shared__ double red[256];
  __shared__ int s_last;
  // the workgroup's threads work out the frame log-likelihoods (four float64 exp and a log each), threads 0 .. 255
  // then add them up in a fixed order: thread t takes frames t, t + 256, ... (the average does not depend on the
  // workgroup size)
  constexpr int FB_LL_LDS = 2048;
  __shared__ float s_ll[FB_LL_LDS];
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
  if (threadIdx.x == 0) {
    const int tvb = r1 - r0;
    double avg = tvb > 0 ? red[0] / (double)tvb : __longlong_as_double(0x7ff8000000000000ll);
    if (g.text_scores && tvb > 0) avg = fb_round6(avg);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(raw + (size_t)b * g.M + m),
                       (unsigned long long)__double_as_longlong(avg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = (atomicAdd(counter, 1) == (int)(gridDim.x * gridDim.y) - 1);
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  fb_loss_body<SMALL, true>(raw, tv, B, g.M, task, 0, attack_type, z_mean, z_std, threshold, adver_thresh, target,
                            true_label, dist_part, n_dist_part, scores, loss, out, ctl, trace, it);
}
void fb_launch_gmm_finalize_loss(hipStream_t s, const FbGmmDev &g, const float *part_m, const float *part_s,
                                 int rows_cap, int n_chunks, const int *row_off, int B, double *raw, int *counter,
                                 const int *tv, int task, int attack_type, const double *z_mean, const double *z_std,
                                 double threshold, double adver_thresh, int target, int true_label,
                                 const double *dist_part, int n_dist_part, double *scores, double *loss, FbNesDev *out,
                                 FbCtlDev *ctl, double *trace, int it) {
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
