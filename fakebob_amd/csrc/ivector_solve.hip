// ivector_solve.hip -- K10c: the B posterior systems (I + sum_k N_k U_k) w = lin of the i-vector extractor
// (`quadratic.Invert()` + mat-vec of IvectorExtractor::GetIvectorDistribution, [EXT] SURVEY.md A.9; the reference
// reaches it through `ivector-extract`, ivector_PLDA_kaldiHelper.py:202-204).
//
// k_iv_solve_ll: one workgroup (8 waves) per utterance, blocked LEFT-looking Cholesky in place on the packed lower
// triangle the contraction wrote, the right-hand side carried along as row R (the forward substitution falls out of
// the panel solves), then a blocked back substitution.
//
// Why left-looking (round 3).  The round-2 kernel was right-looking: every panel read, modified and wrote the whole
// trailing matrix through L2 (10.7 MB of scattered read-modify-write per utterance; 31 us per panel of which the
// matrix cores worked 6) and its 400 x 400 doubles fit neither LDS nor the register file.  Here a panel (32 columns,
// all rows below) is touched ONCE: its tiles are accumulators in registers, S = sum over ALL finished panels q of
// L[rows, q] L[pivot rows, q]^T streamed from the rows written earlier (each lane reads 64 contiguous bytes per row and
// panel: the K index of the MFMA is permuted so that a lane's eight k values are adjacent), then A - S, the diagonal
// block's factorisation, the panel solve against its inverse, one store.  Reads 2.7 MB per utterance, writes 0.64 MB.
//
// The diagonal block (32 x 32) is factored AND inverted by one wave in registers as before (lanes 0..31 = rows of A,
// lanes 32..63 = columns of the identity, same instructions), but the critical path no longer goes through LDS:
// the ONE column the next pivot needs is updated with the multiplier taken by v_readlane from the column just
// scaled; all other columns get that update one pivot later from the LDS broadcast, dealt into the gaps of the next
// pivot's rsq / Newton chain.  13 serial blocks of 32 pivots are the kernel's critical path, so the pivot step is
// what the kernel's time is made of.
#include <float.h>
#include <stdlib.h>

#include <algorithm>

#include "fb_device.h"
#include "fb_kernels.h"
#include "fb_iv_tail.h"

typedef double fb_d4 __attribute__((ext_vector_type(4)));  // accumulator of v_mfma_f64_16x16x4_f64
typedef double fb_d2u __attribute__((ext_vector_type(2), aligned(8)));  // 16-byte load from an 8-byte aligned packed row

#define FB_SV_NB 32   // panel width
// NTR (template parameter of the kernel): 16-row tiles of a panel per wave; waves 1 .. 7 hold them (wave 0 factors):
// 7 x 16 NTR rows >= R + 1, i.e. NTR = 4 up to R = 447 (the recipe's 400), 5 up to R = 512
#define FB_SV_LD 33   // LDS row stride of the 32 x 32 blocks
#define FB_SV_LDC 34  // ... of the per-wave 16 x 32 layout-conversion tile

__device__ __forceinline__ double fb_sv_readlane(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}

// One slice of the deferred updates of column C's iteration: X[cc] -= lprev * col[cc] for a share of cc = C+1 .. 31
// (lprev = the PREVIOUS pivot's scaled column, col = its LDS broadcast), one scheduling region.
template <int C, int S>
__device__ __forceinline__ void fb_sv_gap(double (&X)[FB_SV_NB], const double (&col)[FB_SV_NB], double lprev) {
  if constexpr (C >= 1) {
    constexpr int P = FB_SV_NB - 1 - C, Q = (P + 8) / 9;
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      constexpr int base = C + 1 + S * Q;
      if (base + u < FB_SV_NB) X[base + u] = fma(-lprev, col[base + u], X[base + u]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}
// Pivot C.  On entry X[C] carries every earlier pivot; the columns behind it lack pivot C-1's update, whose scaled
// column lprev every lane finds in the broadcast buffer bc[(C-1) & 1].
template <int C>
__device__ __forceinline__ void fb_sv_col(double (&X)[FB_SV_NB], double *__restrict__ bc, int lane, double lprev, bool &bad) {
  // The pivot and its reciprocal square root FIRST (round 5): the chain of the whole factorisation runs through them, and
  // in front of them stood the requests of the broadcast column (a dozen LDS instructions' issue) and the first slice of
  // the deferred updates, which waits for that column to arrive -- ~200 of a pivot's ~470 cycles (ISA of round 4's build)
  const double d = fb_sv_readlane(X[C], C);
  double ri;                            // 1/sqrt(d): v_rsq_f64 + two Newton steps (full double precision)
  // (as a volatile statement with a memory clobber: left to the compiler the instruction sinks to its first use, behind
  //  the column's requests again)
  asm volatile("v_rsq_f64_e32 %0, %1" : "=v"(ri) : "s"(d) : "memory");
  __builtin_amdgcn_sched_barrier(0);
  double col[FB_SV_NB];
  if constexpr (C >= 1) {
#pragma unroll
    for (int cc = C + 1; cc < FB_SV_NB; ++cc) col[cc] = bc[((C - 1) & 1) * 64 + cc];
  }
  bad |= !(d > 0.0);                // off the chain: a non-positive pivot yields NaNs below and is reported
  // the first Newton step while the column is on its way from LDS (a slice of the deferred updates in front of it would
  // hold the chain until the column arrives); the slices then fill the bubbles of the second step
  const double hd = -0.5 * d;
  double t = hd * ri;
  __builtin_amdgcn_sched_barrier(0);
  double u1 = fma(t, ri, 1.5);
  __builtin_amdgcn_sched_barrier(0);
  ri = ri * u1;
  __builtin_amdgcn_sched_barrier(0);
  fb_sv_gap<C, 0>(X, col, lprev);   // (the first slice holds column C + 1, which the readlane step below continues)
  fb_sv_gap<C, 1>(X, col, lprev);
  fb_sv_gap<C, 2>(X, col, lprev);
  t = hd * ri;
  fb_sv_gap<C, 3>(X, col, lprev);
  fb_sv_gap<C, 4>(X, col, lprev);
  u1 = fma(t, ri, 1.5);
  fb_sv_gap<C, 5>(X, col, lprev);
  fb_sv_gap<C, 6>(X, col, lprev);
  ri = ri * u1;
  fb_sv_gap<C, 7>(X, col, lprev);
  fb_sv_gap<C, 8>(X, col, lprev);
  const double l = X[C] * ri;       // lower half: L[rr][C]; upper half: Linv[C][rr]
  X[C] = l;
  if constexpr (C + 1 < FB_SV_NB) {
    if constexpr (C + 2 < FB_SV_NB) {
      bc[(C & 1) * 64 + lane] = l;  // for the columns behind C + 1, one pivot later; a wave's LDS accesses stay in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const double lc1 = fb_sv_readlane(l, C + 1);  // L[C+1][C]: the only multiplier the next pivot waits for
    X[C + 1] = fma(-l, lc1, X[C + 1]);
    fb_sv_col<C + 1>(X, bc, lane, l, bad);
  }
}
// X: in = A[rr][c] (lower triangle, zeros above; identity rows beyond a short block) in lanes 0..31, delta(r, rr) in
// lanes 32..63; out = L[rr][c] / Linv[r][rr].  bc: 128 doubles of LDS.  Returns true on a non-positive pivot.
__device__ __forceinline__ bool fb_sv_chol32(double (&X)[FB_SV_NB], double *__restrict__ bc, int lane) {
  bool bad = false;
  fb_sv_col<0>(X, bc, lane, 0.0, bad);
  return bad;
}

// S[t] += L[rows of tile t, panel q] L[pivot rows, panel q]^T for q = 0 .. np-1: tiles t < NT of a wave.  pa / pb point at
// this lane's eight k values of panel 0 in its operand rows -- or into a row of zeros where the lane's row does not
// exist, so that nothing is masked after a load and the loop body has no branches.  The k values of a lane are adjacent
// (8 l4 .. 8 l4 + 7 within the panel: any assignment of k to the MFMA's K slots is a valid sum as long as both operands
// use the same one): 64 contiguous bytes per lane, row and panel.
// PIPE: the operands of panel q + 1 are requested before panel q is multiplied (two static register sets, two panels per
// trip, the last request repeated rather than branched around) -- for the late panels, whose few tiles per wave make
// one L2 round trip per panel longer than the panel's MFMAs.
template <int NTR, int NT, bool PIPE>
__device__ __forceinline__ void fb_sv_accumulate(fb_d4 (&S)[NTR][2], const double *const (&pa)[NTR],
                                                 const double *pb0, const double *pb1, int np) {
  auto load8 = [](const double *p, double (&f)[8]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const fb_d2u v = *reinterpret_cast<const fb_d2u *>(p + 2 * h);
      f[2 * h] = v[0];
      f[2 * h + 1] = v[1];
    }
  };
  auto mma = [&](int t, const double (&af)[8], const double (&b0)[8], const double (&b1)[8]) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      S[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s8], b0[s8], S[t][0], 0, 0, 0);
      S[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s8], b1[s8], S[t][1], 0, 0, 0);
    }
  };
  if constexpr (PIPE) {
    double a0[NT][8], a1[NT][8], b00[8], b01[8], b10[8], b11[8];
    load8(pb0, b00);
    load8(pb1, b01);
#pragma unroll
    for (int t = 0; t < NT; ++t) load8(pa[t], a0[t]);
    for (int q = 0; q < np; q += 2) {
      const int k1 = FB_SV_NB * min(q + 1, np - 1), k2 = FB_SV_NB * min(q + 2, np - 1);
      load8(pb0 + k1, b10);
      load8(pb1 + k1, b11);
#pragma unroll
      for (int t = 0; t < NT; ++t) load8(pa[t] + k1, a1[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) mma(t, a0[t], b00, b01);
      if (q + 1 < np) {
        load8(pb0 + k2, b00);
        load8(pb1 + k2, b01);
#pragma unroll
        for (int t = 0; t < NT; ++t) load8(pa[t] + k2, a0[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) mma(t, a1[t], b10, b11);
      }
    }
  } else {
    for (int q = 0; q < np; ++q) {
      const int k0 = FB_SV_NB * q;
      double b0[8], b1[8], af[NT][8];
      load8(pb0 + k0, b0);
      load8(pb1 + k0, b1);
#pragma unroll
      for (int t = 0; t < NT; ++t) load8(pa[t] + k0, af[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) mma(t, af[t], b0, b1);
    }
  }
}

// Back substitution L^T x = y, shared by both kernels (round 4: 45 -> ~16 us).  rhs[0 .. R) (LDS) holds y; block by block
// from the end:  x_p = L_pp^-T t_p  (32 threads, the inverted factor through Di), then RIGHT-looking: every remaining
// column's t loses block row p's contribution at once, t[col] -= sum_r L[32 p + r][col] x_p[r] -- thread = column, the
// block ROW is contiguous in the packed triangle (coalesced), and block row p - 1 is requested before block p's update
// is multiplied, so the serial chain per block is a 32-term triangular product, a 32-term update and three barriers, not
// a memory round trip.  (Round 3 formed t_p LEFT-looking from the column block of all rows below: strided 8-byte reads
// and one exposed L2 round trip per block.)
__device__ __forceinline__ void fb_sv_backsub(const double *__restrict__ Qb, long long aug_off, int R, int npanel,
                                              const double *__restrict__ Lg, double *__restrict__ rhs, double *__restrict__ Di,
                                              double *__restrict__ xs, int tid, int nt) {
  (void)aug_off;
  (void)nt;   // 512 threads: two elements of a 32 x 32 inverse each
  auto rowp = [&](int r) -> const double * { return Qb + (((long long)r * (r + 1)) >> 1); };
  struct Buf { double lv[FB_SV_NB]; double dv[2]; };  // a thread's column of a block row + its two elements of the inverse
  auto fetch = [&](int p, Buf &f) {
    const int j0 = p * FB_SV_NB, nb = min(FB_SV_NB, R - j0);
#pragma unroll
    for (int u = 0; u < 2; ++u) f.dv[u] = Lg[(size_t)p * FB_SV_NB * FB_SV_NB + tid + u * 512];
#pragma unroll
    for (int r = 0; r < FB_SV_NB; ++r) f.lv[r] = (tid < j0 && r < nb) ? rowp(j0 + r)[tid] : 0.0;
  };
  auto step = [&](int p, const Buf &f) {
    const int j0 = p * FB_SV_NB, nb = min(FB_SV_NB, R - j0);
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int idx = tid + u * 512; Di[(idx >> 5) * FB_SV_LD + (idx & 31)] = f.dv[u]; }
    __syncthreads();
    if (tid < 64) {  // x = L_pp^-T t: x[i] = sum_{q >= i} Linv[q][i] t[q]; wave 0, lane = (half of the q range, i).
      // Every LDS operand is requested before the first addition (read inside the loop they were 64 dependent LDS round
      // trips per block: the 3.4 us a step of this substitution took whatever the prefetch depth, r04_rw_stamps_c.txt)
      const int i = tid & 31, hq = tid >> 5;
      double dc[16], tq[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int q = 16 * hq + u;
        dc[u] = Di[q * FB_SV_LD + i];
        tq[u] = rhs[min(j0 + q, R - 1)];
      }
      double x = 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int q = 16 * hq + u;
        x += (q >= i && q < nb) ? dc[u] * tq[u] : 0.0;
      }
      x = x + __shfl_xor(x, 32, 64);   // (lower q half + upper q half: the same sum in both lanes)
      if (tid < FB_SV_NB) xs[tid] = tid < nb ? x : 0.0;
    }
    __syncthreads();
    if (tid < FB_SV_NB && tid < nb) rhs[j0 + tid] = xs[tid];
    if (tid < j0) {
      double xr[FB_SV_NB];
#pragma unroll
      for (int r = 0; r < FB_SV_NB; ++r) xr[r] = xs[r];
      double acc = rhs[tid];
#pragma unroll
      for (int r = 0; r < FB_SV_NB; ++r) acc = fma(-f.lv[r], xr[r], acc);
      rhs[tid] = acc;
    }
    __syncthreads();
  };
  // two register sets, TWO block rows in flight: block row p - 2 is requested when block p's update has consumed its set,
  // a whole step before it is needed
  Buf A, Bb;
  fetch(npanel - 1, A);
  if (npanel >= 2) fetch(npanel - 2, Bb);
  for (int p = npanel - 1; p >= 0; p -= 2) {
    step(p, A);
    if (p - 2 >= 0) fetch(p - 2, A);
    if (p - 1 >= 0) {
      step(p - 1, Bb);
      if (p - 3 >= 0) fetch(p - 3, Bb);
    }
  }
}

template <int NTR>
__global__ __launch_bounds__(512) void k_iv_solve_ll(FbIvDev iv, double *__restrict__ quad, const double *__restrict__ linp,
                                                     int n_kchunks, int B, double *__restrict__ AugAll,
                                                     double *__restrict__ LinvAll, double *__restrict__ ivec,
                                                     int *__restrict__ fail, FbIvTail tail) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  const int R = iv.R, b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  // tile row tr of a panel belongs to wave 1 + tr % 7 (its tile number tr / 7); wave 0 keeps its registers for the
  // factorisation
  const int tw = wv - 1;
  auto tile_row = [&](int t) { return wv == 0 ? 0x7fff : tw + 7 * t; };
  const int npanel = (R + FB_SV_NB - 1) / FB_SV_NB;
  double *Q = quad + (size_t)b * iv.triR;
  double *aug = AugAll + (size_t)b * R;
  // Rows of the matrix (packed), the right-hand side as row R, and a row of zeros (R + 64 of them behind the
  // right-hand sides, kept by the engine) for the lanes whose row or element does not exist: all as element offsets
  // from Q, chosen with integer selects -- nothing is masked after a load and the loops have no divergent branches.
  const long long aug_off = (long long)(aug - Q), zero_off = (long long)((AugAll + (size_t)B * R) - Q);
  auto roff = [&](int r) -> long long { return r < R ? (long long)((r * (r + 1)) >> 1) : aug_off; };  // R <= 2^15
  double *Lg = LinvAll + (size_t)b * npanel * FB_SV_NB * FB_SV_NB;
  double *rhs = smd;                          // [R]
  double *Dg = rhs + ((R + 1) & ~1);          // [32][33] the diagonal block on its way to wave 0; later scratch
  double *Di = Dg + FB_SV_NB * FB_SV_LD;      // [32][33] its inverse
  double *bc = Di + FB_SV_NB * FB_SV_LD;      // [128] column broadcast of the factorisation
  double *cvt = bc + 128;                     // [8 waves][16][34] accumulator layout -> A-operand layout
  for (int r = tid; r < R; r += nt) Q[((r * (r + 1)) >> 1) + r] += 1.0;  // A = I + quad
  // rhs = sum of the lin partials: 8 interleaved slices per component, combined in fixed order (thread = component:
  // coalesced, 8 independent loads in flight)
  for (int r = tid; r < R; r += nt) {
    double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const double *lp = linp + (size_t)b * R + r;
    int ch = 0;
    for (; ch + 8 <= n_kchunks; ch += 8) {
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) a8[sl] += lp[(size_t)(ch + sl) * B * R];
    }
    for (int sl = 0; ch + sl < n_kchunks; ++sl) a8[sl] += lp[(size_t)(ch + sl) * B * R];
    double acc = 0.0;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) acc += a8[sl];
    aug[r] = acc + (r == 0 ? iv.prior_offset : 0.0);
  }
  __syncthreads();

  for (int j0 = 0, pi = 0; j0 < R; j0 += FB_SV_NB, ++pi) {
    const int nb = min(FB_SV_NB, R - j0);
    const int ntr = (R + 1 - j0 + 15) >> 4;  // 16-row tiles of the panel: the diagonal block, the rows below, the rhs row
    // ---- phase 1: S = sum over the finished panels q of L[rows, q] L[pivot rows, q]^T - A, tiles in registers.  The
    //      accumulators START from -A (element i of a lane = row 4 i + l4, column l15 of the tile): those loads are in
    //      flight together with the first operands instead of costing their own L2 round trip afterwards.
    fb_d4 S[NTR][2];
    int nt_w = 0;  // valid tiles of this wave (wave-uniform)
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      const int tr = tile_row(t);
      if (tr < ntr) nt_w = t + 1;
      double av[2][4];
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = j0 + 16 * tr + l4 + 4 * i, cc = j0 + 16 * tc + l15;
          const bool ok = tr < ntr && rr <= R && cc < j0 + nb && cc <= rr;
          av[tc][i] = Q[ok ? roff(rr) + cc : zero_off];
        }
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int i = 0; i < 4; ++i) S[t][tc][i] = -av[tc][i];
    }
    nt_w = __builtin_amdgcn_readfirstlane(nt_w);
    if (pi > 0 && nt_w > 0) {
      // pivot rows of this lane's B fragments (column tiles 0 / 1): rows j0 + l15, j0 + 16 + l15 of the matrix proper
      const double *pb0 = Q + (l15 < nb ? roff(j0 + l15) : zero_off) + 8 * l4,
                   *pb1 = Q + (16 + l15 < nb ? roff(j0 + 16 + l15) : zero_off) + 8 * l4;
      const double *pa[NTR];
#pragma unroll
      for (int t = 0; t < NTR; ++t) {
        const int ra = j0 + 16 * tile_row(t) + l15;
        pa[t] = Q + ((tile_row(t) < ntr && ra <= R) ? roff(ra) : zero_off) + 8 * l4;
      }
      switch (nt_w) {
        case 1: fb_sv_accumulate<NTR, 1, true>(S, pa, pb0, pb1, pi); break;
        case 2: fb_sv_accumulate<NTR, 2, true>(S, pa, pb0, pb1, pi); break;
        case 3: fb_sv_accumulate<NTR, 3, false>(S, pa, pb0, pb1, pi); break;
        case 4: fb_sv_accumulate<NTR, 4, false>(S, pa, pb0, pb1, pi); break;
        default: fb_sv_accumulate<NTR, NTR, false>(S, pa, pb0, pb1, pi); break;
      }
    }
    // ---- A - S: the sign; a diagonal tile's upper part saw real operands and is zeroed here
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      const int tr = tile_row(t);
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = j0 + 16 * tr + l4 + 4 * i, cc = j0 + 16 * tc + l15;
          const bool ok = tr < ntr && rr <= R && cc < j0 + nb && cc <= rr;
          S[t][tc][i] = ok ? -S[t][tc][i] : 0.0;
        }
    }
    if (wv == 1 || wv == 2) {  // tile rows 0 and 1 are the diagonal block (tile 0 of waves 1 and 2): to LDS for the factorisation
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int i = 0; i < 4; ++i) Dg[(16 * tw + l4 + 4 * i) * FB_SV_LD + 16 * tc + l15] = S[0][tc][i];
    }
    __syncthreads();
    // ---- phase 2: wave 0 factors the block and inverts the factor (registers only); L11 -> A, L11^-1 -> Di, Lg
    if (wv == 0) {
      double X[FB_SV_NB];
      // (rr passes through an empty asm: the 32 identity values (rr == c ? 1 : 0) below do not change from panel to
      //  panel, so hipcc computed them once in front of the panel loop and -- 64 registers that cannot stay live across
      //  it -- spilled them: the kernels' 65 / 104 spilled registers of round 4.  Two instructions each, made in place.)
      int rr = lane & 31;
      asm volatile("" : "+v"(rr));
      if (lane < FB_SV_NB) {
#pragma unroll
        for (int c = 0; c < FB_SV_NB; ++c) {
          const double v = Dg[rr * FB_SV_LD + c];
          X[c] = (rr < nb && c < nb) ? (c <= rr ? v : 0.0) : (rr == c ? 1.0 : 0.0);
        }
      } else {
#pragma unroll
        for (int c = 0; c < FB_SV_NB; ++c) X[c] = (rr == c) ? 1.0 : 0.0;
      }
      const bool bad = fb_sv_chol32(X, bc, lane);
      if (bad && lane == 0) atomicMax(fail, b + 1);
      if (lane < FB_SV_NB) {
#pragma unroll
        for (int c = 0; c < FB_SV_NB; ++c)
          if (rr < nb && c <= rr) Q[roff(j0 + rr) + j0 + c] = X[c];
      } else {
#pragma unroll
        for (int r = 0; r < FB_SV_NB; ++r) {
          Di[r * FB_SV_LD + rr] = X[r];
          Lg[((size_t)pi * FB_SV_NB + r) * FB_SV_NB + rr] = X[r];
        }
      }
    }
    __syncthreads();
    // ---- phase 3: the rows below, X = (A - S) L11^-T on the matrix cores; the tile changes from the accumulator layout
    //      to the A-operand layout through a per-wave LDS tile; one store per element of the panel
    {
      double *cw = cvt + (size_t)wv * 16 * FB_SV_LDC;
      double b0[8], b1[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {  // B fragments: Linv[column][k], k = 4 s + l4 (zeros above the diagonal are stored)
        b0[s8] = Di[l15 * FB_SV_LD + 4 * s8 + l4];
        b1[s8] = Di[(16 + l15) * FB_SV_LD + 4 * s8 + l4];
      }
#pragma unroll
      for (int t = 0; t < NTR; ++t) {
        const int tr = tile_row(t);
        // rows behind the diagonal block: tiles 2 .. when the block is full; a short last block (nb < 32) has only the
        // rhs row behind it, in tile nb / 16, next to block rows that are masked at the store
        if (tr >= (nb >> 4) && tr < ntr) {  // wave-uniform
#pragma unroll
          for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int i = 0; i < 4; ++i) cw[(l4 + 4 * i) * FB_SV_LDC + 16 * tc + l15] = S[t][tc][i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          fb_d4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) {
            const double av = cw[l15 * FB_SV_LDC + 4 * s8 + l4];
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0[s8], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1[s8], x1, 0, 0, 0);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();  // the tile is read before the next one overwrites it
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = j0 + 16 * tr + l4 + 4 * i;
            if (rr <= R && rr >= j0 + nb) {
              double *ao = Q + roff(rr) + j0;
              if (l15 < nb) ao[l15] = x0[i];
              if (16 + l15 < nb) ao[16 + l15] = x1[i];
            }
          }
        }
      }
    }
    __syncthreads();  // the panel is in memory before the next one streams it
  }
  // ---- the augmented row now holds y = L^-1 rhs; L^T x = y block by block from the end: for block p
  //      x_p = L_pp^-T (y_p - sum over the rows k below of L[k][p columns] x_k), every row of L read once, 256
  //      contiguous bytes at a time (thread = (row group of 16, column))
  for (int r = tid; r < R; r += nt) rhs[r] = aug[r];
  __syncthreads();
  fb_sv_backsub(Q, aug_off, R, npanel, Lg, rhs, Di, Dg, tid, nt);
  // the i-vector row, then -- in this workgroup, which holds the solution -- the back-end, and in the workgroup that
  // finishes last the loss body (fb_iv_tail.h); everything behind rhs is free now
  fb_iv_tail_run<512>(iv, tail, b, B, rhs, ivec, Dg);
}

// ---------------------------------------------------------------------------------------------------------------
// k_iv_solve_rw (round 4): the same factorisation ROW-WISE over G (= FB_RW_G = 5) workgroups per matrix.  One workgroup per matrix has a
// floor of ~200 us -- 13 serial 32 x 32 block factorisations of 7 us with seven waves idle, and at samples_per_draw = 50
// only 51 of 256 compute units have a matrix at all.  Here block row rb (32 rows: tile rows 2 rb, 2 rb + 1; the right-hand
// side rides along as row R) belongs to workgroup rb % G, which does everything for it, column by column ("up-looking"
// Cholesky):
//     for c < rb:  S = A[rb, c] - sum_{q < c} L[rb, q] L[c, q]^T      (needs L[c, 0 .. c-1]: row c's off-diagonals)
//                  L[rb, c] = S L_cc^-T                                (needs row c COMPLETE: its inverted factor)
//     diagonal:    S = A[rb, rb] + I - sum_{q < rb} L[rb, q] L[rb, q]^T, factor + invert (wave 0, fb_sv_chol32), publish
// Block rows complete strictly in order, but when row rb - 1 completes the owner of row rb has everything else behind
// it: what is left is ONE panel solve, one 32 x 32 x 32 update of its diagonal block (the earlier contributions are
// summed beforehand) and its own factorisation -- the serial chain is hop + solve + update + factor per row, and the
// bulk (the sums over q for the next rows) runs on the other workgroups of the matrix meanwhile.
// Cross-workgroup traffic without device-wide fences (a release / acquire pair writes back / invalidates an XCD's whole
// L2 on the eight-XCD MI355X: microseconds each, k_vad_delta_cmvn_p's lesson): every L block and inverse is written with
// agent-scope (write-through) stores and read with agent-scope loads;
//   * prog[b][rb] = (epoch << 8) | n: block row rb has its first n column blocks in memory (n = rb: off-diagonals done,
//     n = rb + 1: complete).  Written by one thread after a workgroup barrier behind s_waitcnt vmcnt(0) of every storing
//     thread -- the stores are complete, i.e. visible at agent scope, before the flag is issued;
//   * the inverted diagonal factors are polled as DATA: two slot sets alternate with the launch epoch, a slot holding the
//     sentinel NaN has not been written (the owner puts the sentinel back into the other set's slots for the launch after
//     next) -- the hop of the serial chain is one memory round trip, not flag-then-data.
// Workgroups draw (matrix, g) from a ticket, g fastest: every smaller ticket is running or done; a workgroup waits only
// for block rows below its own, and a matrix whose G workgroups have all started needs nobody else -- at most one matrix
// is partly started at any time, so waiting cannot deadlock.
#define FB_RW_SENT 0x7ff87ff87ff87ff8ull
#ifdef FB_RW_STAMP  // instrumented build (tools/profile/rw_instrumented.sh): the serial chain of matrix 0, per block row
__device__ unsigned long long g_rw_stamps[16 * 8 + 8];
extern "C" int fb_debug_rw_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rw_stamps), sizeof(g_rw_stamps)) == hipSuccess ? 0 : -1;
}
#define RW_STAMP(row, k) do { if (b == 0 && tid == 0) g_rw_stamps[(row) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define RW_STAMP(row, k) do { } while (0)
#endif
__device__ __forceinline__ double fb_rw_ld(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), FB_XCH_LD,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void fb_rw_st(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), FB_XCH_ST,
                     __HIP_MEMORY_SCOPE_AGENT);
}
template <int G>
__global__ __launch_bounds__(512) void k_iv_solve_rw(FbIvDev iv, double *__restrict__ quad, const double *__restrict__ linp,
                                                     int n_kchunks, int B, double *__restrict__ AugAll,
                                                     double *__restrict__ LinvAll, double *__restrict__ ivec,
                                                     int *__restrict__ fail, unsigned *__restrict__ prog,
                                                     int *__restrict__ ticket, unsigned epoch, FbIvTail tail) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  __shared__ int s_tk;
  const int R = iv.R, tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  if (tid == 0) {
    const int tk = atomicAdd(ticket, 1);
    if (tk == B * G - 1) __hip_atomic_store(ticket, 0, FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    s_tk = tk;
  }
  __syncthreads();
  const int b = s_tk / G, g = s_tk - b * G;
  const int nbr = (R + 1 + FB_SV_NB - 1) / FB_SV_NB;   // block rows incl. the right-hand side (row R)
  const int npanel = (R + FB_SV_NB - 1) / FB_SV_NB;    // diagonal blocks
  double *Q = quad + (size_t)b * iv.triR;
  double *aug = AugAll + (size_t)b * R;
  const long long aug_off = (long long)(aug - Q), zero_off = (long long)((AugAll + (size_t)B * R) - Q);
  auto roff = [&](int r) -> long long { return r < R ? (long long)((r * (r + 1)) >> 1) : aug_off; };
  const size_t lset = (size_t)B * npanel * FB_SV_NB * FB_SV_NB;
  double *Lcur = LinvAll + (size_t)(epoch & 1u) * lset + (size_t)b * npanel * FB_SV_NB * FB_SV_NB;
  double *Lnxt = LinvAll + (size_t)((epoch + 1u) & 1u) * lset + (size_t)b * npanel * FB_SV_NB * FB_SV_NB;
  unsigned *pg = prog + (size_t)b * (nbr + 1);
  // LDS: the own block row as it is solved (32 rows x 32 npanel columns: the A operand of every later sum is read from
  // here, never from memory), the block being formed, an inverted diagonal factor, exchange areas.  The back
  // substitution at the end re-uses the row area.
  const int ldr = FB_SV_NB * npanel + 2;        // row stride of Lrow (doubles)
  double *Lrow = smd;                          // [32][ldr]
  double *Di = Lrow + (size_t)FB_SV_NB * ldr;  // [32][33] an inverted diagonal factor
  double *bc = Di + FB_SV_NB * FB_SV_LD;       // [128] column broadcast of the factorisation
  double *Sx = bc + 128;                       // [32][34] the block being formed, row-major (A-operand layout of the solve)
  double *pr = Sx + FB_SV_NB * FB_SV_LDC;      // [4 sub-tiles][16][17] the second K half's partial sums
  double *rhs = Lrow;                          // back substitution: [R], then Dg [32], red [16][33]
  double *Dg = rhs + ((R + 1) & ~1);
  __shared__ int s_cnt, s_ready;               // stores-complete count of the solving waves (off-diagonal flag); complete block rows seen
  if (tid == 0) s_cnt = 0;
  const int st = wv & 3, th = st >> 1, tc = st & 1, kh = wv >> 2;   // wave -> 16 x 16 sub-tile (th, tc), K half kh

  auto wait_prog = [&](int row, unsigned need) {  // one thread polls, the barrier releases the workgroup
    if (tid == 0) {
      unsigned v;
      do {
        v = __hip_atomic_load(&pg[row], FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
      } while ((v >> 8) != (epoch & 0xffffffu) || (v & 0xffu) < need);
    }
    __syncthreads();
  };
  auto publish = [&](int row, unsigned n) {       // every thread's stores are complete, then the flag
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_max(&pg[row], ((epoch & 0xffffffu) << 8) | n, FB_XCH_RMW, __HIP_MEMORY_SCOPE_AGENT);
  };
  // acc(sub-tile st) = sum over the panels q = kh, kh + 2, .. < nq of Lrow[rows 16 th .., q] L[rows c0 + 16 tc .., q]^T:
  // the own rows from LDS, the other block row's from memory -- ALL its fragments requested before the first multiply
  // (one memory round trip, whatever nq: the serial chain of the factorisation waits for this sum once per block row)
  constexpr int FB_RW_MAXP = 7;                 // panels per K half: R <= 448
  auto accumulate = [&](int c0, int nq) -> fb_d4 {
    fb_d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int rbb = c0 + 16 * tc + l15;
    const double *pb = Q + (rbb < R ? roff(rbb) : zero_off) + 8 * l4;
    const double *la = Lrow + (size_t)(16 * th + l15) * ldr + 8 * l4;
    // PLAIN (L2-cached) 16-byte loads: this XCD cannot hold a stale line of these blocks -- nobody reads block row c's
    // off-diagonal part before its flag, the writer's stores went through to memory before the flag, and a line that
    // straddles into the diagonal block's not yet final part is never read through the cache for that part.  (As
    // agent-scope loads every block came from memory every time: the catch-up columns of the late block rows took
    // 5 - 11 us each and the row-wise kernel fell behind its own serial chain, profiles/r04_rw_stamps_a.txt.)
    double bf[FB_RW_MAXP][8];
#pragma unroll
    for (int j = 0; j < FB_RW_MAXP; ++j) {
      const int q = kh + 2 * j;
      if (q < nq) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const fb_d2u v = *reinterpret_cast<const fb_d2u *>(pb + FB_SV_NB * q + 2 * h);
          bf[j][2 * h] = v[0];
          bf[j][2 * h + 1] = v[1];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < FB_RW_MAXP; ++j) {
      const int q = kh + 2 * j;
      if (q < nq) {
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[FB_SV_NB * q + s8], bf[j][s8], acc, 0, 0, 0);
      }
    }
    return acc;
  };
  // this lane's elements of A[block at (r0, c0)] (+ I on the diagonal), requested ahead of their use
  auto load_a = [&](int r0, int c0, double (&av)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = r0 + 16 * th + l4 + 4 * i, cc = c0 + 16 * tc + l15;
      const bool ok = rr <= R && cc < R && cc <= rr;
      av[i] = fb_rw_ld(Q + (ok ? roff(rr) + cc : zero_off)) + ((ok && rr == cc) ? 1.0 : 0.0);
    }
  };
  // Sx = A block - (acc of K half 0 + acc of K half 1), zero where the block has no element
  auto finish_block = [&](const fb_d4 &acc, const double (&av)[4], int r0, int c0, bool diag) {
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[(st * 16 + l4 + 4 * i) * 17 + l15] = acc[i];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = r0 + 16 * th + l4 + 4 * i, cc = c0 + 16 * tc + l15;
        const bool ok = rr <= R && cc < R && (!diag || cc <= rr);
        const double v = av[i] - (acc[i] + pr[(st * 16 + l4 + 4 * i) * 17 + l15]);
        Sx[(16 * th + l4 + 4 * i) * FB_SV_LDC + 16 * tc + l15] = ok ? v : 0.0;
      }
    }
    __syncthreads();
  };
  // the right-hand side (sum of the lin partials) is prepared by the workgroup with the longest wait before its first
  // row, g = G - 1, and announced as prog[nbr]
  if (g == G - 1) {
    for (int r = tid; r < R; r += nt) {
      double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const double *lp = linp + (size_t)b * R + r;
      int ch = 0;
      for (; ch + 8 <= n_kchunks; ch += 8) {
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) a8[sl] += lp[(size_t)(ch + sl) * B * R];
      }
      for (int sl = 0; ch + sl < n_kchunks; ++sl) a8[sl] += lp[(size_t)(ch + sl) * B * R];
      double acc = 0.0;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) acc += a8[sl];
      fb_rw_st(&aug[r], acc + (r == 0 ? iv.prior_offset : 0.0));
    }
    publish(nbr, 1u);
  }

  int n_pub = 0;   // off-diagonal flags this workgroup has published (the solving waves count their completed stores)
  for (int rb = g; rb < nbr; rb += G) {
    const int r0 = FB_SV_NB * rb;
    const bool has_diag = r0 < R;
    const int nb = has_diag ? min(FB_SV_NB, R - r0) : 0;
    const int ncol = min(rb, npanel);                    // off-diagonal column blocks of this block row
    RW_STAMP(rb, 0);
    if (r0 + FB_SV_NB > R) wait_prog(nbr, 1u);          // this block row holds the right-hand side
    if (has_diag) {  // the other slot set's inverse of this row: sentinel again for the launch after next
      for (int i = tid; i < FB_SV_NB * FB_SV_NB; i += nt)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(Lnxt + (size_t)rb * FB_SV_NB * FB_SV_NB + i), FB_RW_SENT,
                           FB_XCH_ST, __HIP_MEMORY_SCOPE_AGENT);
    }
    fb_d4 dacc = {0.0, 0.0, 0.0, 0.0};   // this wave's share of sum_q L[rb,q] L[rb,q]^T (sub-tile st, K half kh of every block)
    double av[4], avd[4] = {0.0, 0.0, 0.0, 0.0};
    if (kh == 0) {
      if (has_diag) load_a(r0, r0, avd);
      if (ncol > 0) load_a(r0, 0, av);
    }
    // how many of the block rows below are COMPLETE already: one round trip for all their progress words instead of two
    // per column (a workgroup that comes back from its previous block row has most columns waiting for it; the polls of
    // flags that have long been up were most of a catch-up column's 5 us, profiles/r04_rw_stamps_b.txt)
    if (tid == 0) {
      unsigned pv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) pv[i] = i < ncol ? __hip_atomic_load(&pg[i], FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      int ready = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i == ready && i < ncol && (pv[i] >> 8) == (epoch & 0xffffffu) && (pv[i] & 0xffu) >= (unsigned)(i + 1)) ready = i + 1;
      s_ready = ready;
    }
    __syncthreads();
    const int ready = s_ready;
    auto ld_inv = [&](int c, unsigned long long (&dst)[2]) {   // this thread's two words of row c's inverted factor
#pragma unroll
      for (int u = 0; u < 2; ++u)
        dst[u] = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(Lcur + (size_t)c * FB_SV_NB * FB_SV_NB + tid + u * 512),
                                   FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
    };
    unsigned long long dvn[2] = {FB_RW_SENT, FB_RW_SENT};
    if (ncol > 0 && 0 < ready) ld_inv(0, dvn);
    for (int c = 0; c < ncol; ++c) {
      if (c > 0 && c >= ready) wait_prog(c, (unsigned)c);              // L[c, 0 .. c-1] in memory
      if (c == ncol - 1) RW_STAMP(rb, 1);
      unsigned long long dvc[2] = {dvn[0], dvn[1]};
      if (c + 1 < ncol && c + 1 < ready) ld_inv(c + 1, dvn);           // the next column's inverse: in flight during this one
      const fb_d4 acc = accumulate(FB_SV_NB * c, c);
      finish_block(acc, av, r0, FB_SV_NB * c, false);    // Sx = A[rb, c] - sum_{q < c} ...
      if (c == ncol - 1) RW_STAMP(rb, 2);
      if (kh == 0 && c + 1 < ncol) load_a(r0, FB_SV_NB * (c + 1), av);
      // the inverse of row c's factor: prefetched when row c was known to be complete, else polled as data
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 512;
        unsigned long long bits = c < ready ? dvc[u] : FB_RW_SENT;
        while (bits == FB_RW_SENT)
          bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(Lcur + (size_t)c * FB_SV_NB * FB_SV_NB + i),
                                   FB_XCH_LD, __HIP_MEMORY_SCOPE_AGENT);
        Di[(i >> 5) * FB_SV_LD + (i & 31)] = __longlong_as_double((long long)bits);
      }
      __syncthreads();
      if (c == ncol - 1) RW_STAMP(rb, 3);
      // X = Sx Linv^T on the matrix cores: waves 4 .. 7 (wave 0 never has stores in flight when it starts a
      // factorisation), sub-tile (th, tc); to the LDS row and -- without waiting -- to memory
      if (kh == 1) {
        fb_d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
          x = __builtin_amdgcn_mfma_f64_16x16x4f64(Sx[(16 * th + l15) * FB_SV_LDC + 4 * s8 + l4],
                                                   Di[(16 * tc + l15) * FB_SV_LD + 4 * s8 + l4], x, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rl = 16 * th + l4 + 4 * i, rr = r0 + rl, cc = FB_SV_NB * c + 16 * tc + l15;
          const bool ok = rr <= R && cc < R;
          if (ok) fb_rw_st(Q + roff(rr) + cc, x[i]);
          Lrow[(size_t)rl * ldr + cc] = ok ? x[i] : 0.0;
        }
      }
      __syncthreads();
      // the block's contribution to the diagonal block's sum, while it is hot: K half kh = columns 16 kh .. 16 kh + 15
      if (has_diag) {
        const double *la = Lrow + (size_t)(16 * th + l15) * ldr + FB_SV_NB * c + 16 * kh + l4;
        const double *lb = Lrow + (size_t)(16 * tc + l15) * ldr + FB_SV_NB * c + 16 * kh + l4;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[4 * s4], lb[4 * s4], dacc, 0, 0, 0);
      }
    }
    if (has_diag) finish_block(dacc, avd, r0, r0, true);   // Sx = A[rb,rb] + I - sum_{q < rb} L[rb,q] L[rb,q]^T
    RW_STAMP(rb, 4);
    // "off-diagonals done": the solving waves wait for their stores and count; one of them raises the flag -- no workgroup
    // barrier, wave 0 is factoring meanwhile
    if (ncol > 0 && kh == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) atomicAdd(&s_cnt, 1);
      if (wv == 4 && lane == 0) {
        n_pub += 1;
        while (__hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * n_pub) { }
        // (fetch_max: wave 0's "complete" below may overtake this one; within an epoch the word only grows)
        __hip_atomic_fetch_max(&pg[rb], ((epoch & 0xffffffu) << 8) | (unsigned)ncol, FB_XCH_RMW, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (has_diag) {
      if (wv == 0) {
        double X[FB_SV_NB];
        int rr = lane & 31;              // (through an empty asm: see k_iv_solve_ll -- keeps the identity values out of
        asm volatile("" : "+v"(rr));     //  the block-row loop's preheader, where they were spilled)
        if (lane < FB_SV_NB) {
#pragma unroll
          for (int cc = 0; cc < FB_SV_NB; ++cc) {
            const double v = Sx[rr * FB_SV_LDC + cc];
            X[cc] = (rr < nb && cc < nb) ? (cc <= rr ? v : 0.0) : (rr == cc ? 1.0 : 0.0);
          }
        } else {
#pragma unroll
          for (int cc = 0; cc < FB_SV_NB; ++cc) X[cc] = (rr == cc) ? 1.0 : 0.0;
        }
        const bool bad = fb_sv_chol32(X, bc, lane);
        if (bad && lane == 0) atomicMax(fail, b + 1);
        if (lane >= FB_SV_NB) {   // the inverse first: it is what the next block row's owner is polling for
#pragma unroll
          for (int r = 0; r < FB_SV_NB; ++r) {
            Di[r * FB_SV_LD + rr] = X[r];
            double v = X[r];
            if ((unsigned long long)__double_as_longlong(v) == FB_RW_SENT) v = __longlong_as_double(0x7ff8000000000001ll);
            fb_rw_st(Lcur + ((size_t)rb * FB_SV_NB + r) * FB_SV_NB + rr, v);
          }
        } else {
#pragma unroll
          for (int cc = 0; cc < FB_SV_NB; ++cc)
            if (rr < nb && cc <= rr) fb_rw_st(Q + roff(r0 + rr) + r0 + cc, X[cc]);
        }
        // "complete": a workgroup that starts a later block row reads this word to know that it need not poll for this
        // row at all (its inverse can be requested ahead)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0)
          __hip_atomic_fetch_max(&pg[rb], ((epoch & 0xffffffu) << 8) | (unsigned)(rb + 1), FB_XCH_RMW, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      RW_STAMP(rb, 5);
      // rows of this block row behind a short diagonal block (only the right-hand side can be there): X = S L^-T
      if (r0 + nb <= R && r0 + FB_SV_NB > R && kh == 1) {
        fb_d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
          x = __builtin_amdgcn_mfma_f64_16x16x4f64(Sx[(16 * th + l15) * FB_SV_LDC + 4 * s8 + l4],
                                                   Di[(16 * tc + l15) * FB_SV_LD + 4 * s8 + l4], x, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = r0 + 16 * th + l4 + 4 * i, cc = r0 + 16 * tc + l15;
          if (rr <= R && rr >= r0 + nb && cc < r0 + nb) fb_rw_st(Q + roff(rr) + cc, x[i]);
        }
      }
    }
    if ((nbr - 1) % G == g && rb + G >= nbr) publish(rb, (unsigned)(ncol + (has_diag ? 1 : 0)));   // the last row of the owner of the back substitution: its stores complete
    else __syncthreads();
  }
  if ((nbr - 1) % G != g) return;   // the owner of the last block row goes on with the back substitution
  // everything the other workgroups wrote went through to memory before they published; one invalidate and plain loads do
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  RW_STAMP(15, 0);
  for (int r = tid; r < R; r += nt) rhs[r] = aug[r];
  __syncthreads();
  fb_sv_backsub(Q, aug_off, R, npanel, Lcur, rhs, Di, Dg, tid, nt);
  RW_STAMP(15, 1);
  // the i-vector row, the back-end of this utterance and -- in the matrix owner that finishes last -- the loss body
  // (fb_iv_tail.h), in the row area behind what the back substitution used
  fb_iv_tail_run<512>(iv, tail, b, B, rhs, ivec, rhs + (((R + 1) & ~1) + FB_SV_NB + 16 * FB_SV_LD));
  RW_STAMP(15, 2);
}

// LinvAll: TWO slot sets of B x npanel x 32 x 32 doubles, every word FB_RW_SENT before the first launch / after an epoch
// restart; prog: B x (block rows + 1) unsigned, zero then; ticket: one int, zero.  Returns false when the grid of B x G
// workgroups would not be resident at once (the caller then runs k_iv_solve_ll).
// 5 workgroups per matrix: 255 of the 256 compute units at samples_per_draw = 50 (51 systems).  With 4 the late block rows'
// catch-up (all the columns that became available while the workgroup finished its previous row, ~5 us each) fell behind
// the serial chain from row 7 on: 208 us against 194 (profiles/r04_rw_stamps_d.txt, _e.txt)
#define FB_RW_G 5
size_t fb_iv_solve_rw_linv_doubles(const FbIvDev &iv, int B) { return (size_t)2 * B * ((iv.R + 31) / 32) * 1024; }
size_t fb_iv_solve_rw_prog_words(const FbIvDev &iv, int B) { return (size_t)B * ((iv.R + 1 + 31) / 32 + 1); }
template <int G>
static bool launch_solve_rw(hipStream_t s, size_t shm, const FbIvDev &iv, const double *quad, const double *linp, int n_kchunks,
                            int B, double *Aall, double *LinvAll, double *ivec, int *fail, unsigned *prog, int *ticket,
                            unsigned epoch, const FbIvTail &tail) {
  static std::atomic<unsigned long long> optin{0};
  unsigned long long bit = 0;
  if (fb_device_needs_optin(optin, &bit)) {  // one workgroup per compute unit: more than half of its LDS
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_iv_solve_rw<G>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
      return false;
    optin.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(k_iv_solve_rw<G>, dim3(B * G), dim3(512), std::max(shm, (size_t)82 * 1024), s, iv,
                     const_cast<double *>(quad), linp, n_kchunks, B, Aall, LinvAll, ivec, fail, prog, ticket, epoch, tail);
  return true;
}
// workgroups per matrix: FB_RW_G, fewer when B x FB_RW_G workgroups would not be resident at once (FB_IV_RW_G = 2 | 3 | 5
// chooses for A/B runs)
int fb_iv_solve_rw_groups(int B) {
  int g = FB_RW_G;
  if (const char *ev = getenv("FB_IV_RW_G")) {
    const int v = atoi(ev);
    if (v == 2 || v == 3 || v == 5) g = v;
  }
  while (g > 2 && B * g > 256) g = g == 5 ? 3 : 2;
  return g;
}
bool fb_launch_iv_solve_rw(hipStream_t s, const FbIvDev &iv, const double *quad, const double *linp, int n_kchunks, int B,
                           double *Aall, double *LinvAll, double *ivec, int *fail, unsigned *prog, int *ticket, unsigned epoch,
                           const FbIvTail &tail) {
  const int R = iv.R, G = fb_iv_solve_rw_groups(B);
  if (B * G > 256 || R < 64) return false;
  const int npanel = (R + FB_SV_NB - 1) / FB_SV_NB;
  const size_t rowd = (size_t)FB_SV_NB * (FB_SV_NB * npanel + 2), bsd = ((R + 1) & ~1) + FB_SV_NB + 16 * FB_SV_LD;
  const size_t tld = tail.backend ? bsd + fb_iv_tail_lds_doubles(iv) : 0;   // the tail works behind the back substitution's area
  const size_t shm = sizeof(double) * (std::max(std::max(rowd, bsd), tld) + FB_SV_NB * FB_SV_LD + 128 + FB_SV_NB * FB_SV_LDC + 4 * 16 * 17);
  if (shm > 150 * 1024 || R > 448) return false;
  switch (G) {
    case 2: return launch_solve_rw<2>(s, shm, iv, quad, linp, n_kchunks, B, Aall, LinvAll, ivec, fail, prog, ticket, epoch, tail);
    case 3: return launch_solve_rw<3>(s, shm, iv, quad, linp, n_kchunks, B, Aall, LinvAll, ivec, fail, prog, ticket, epoch, tail);
    default: return launch_solve_rw<5>(s, shm, iv, quad, linp, n_kchunks, B, Aall, LinvAll, ivec, fail, prog, ticket, epoch, tail);
  }
}

template <int NTR>
static void launch_solve_ll(hipStream_t s, size_t shm, const FbIvDev &iv, const double *quad, const double *linp, int n_kchunks,
                            int B, double *Aall, double *LinvAll, double *ivec, int *fail, const FbIvTail &tail) {
  static std::atomic<unsigned long long> optin{0};
  unsigned long long bit = 0;
  if (shm > 64 * 1024 && fb_device_needs_optin(optin, &bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_iv_solve_ll<NTR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess)
      optin.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(k_iv_solve_ll<NTR>, dim3(B), dim3(512), shm, s, iv, const_cast<double *>(quad), linp, n_kchunks, B, Aall,
                     LinvAll, ivec, fail, tail);
}
void fb_launch_iv_solve_ll(hipStream_t s, const FbIvDev &iv, const double *quad, const double *linp, int n_kchunks,
                           int B, double *Aall, double *LinvAll, double *ivec, int *fail, const FbIvTail &tail) {
  const int R = iv.R;
  const size_t own = ((R + 1) & ~1) + 2 * FB_SV_NB * FB_SV_LD + 128 + 8 * 16 * FB_SV_LDC + 16 * FB_SV_LD;
  const size_t tld = tail.backend ? ((R + 1) & ~1) + fb_iv_tail_lds_doubles(iv) : 0;   // the tail re-uses everything behind rhs
  const size_t shm = sizeof(double) * std::max(own, tld);
  if (R + 1 <= 7 * 16 * 4) launch_solve_ll<4>(s, shm, iv, quad, linp, n_kchunks, B, Aall, LinvAll, ivec, fail, tail);
  else launch_solve_ll<5>(s, shm, iv, quad, linp, n_kchunks, B, Aall, LinvAll, ivec, fail, tail);
}
