// gmm_wide_kernel.hip -- k_gmm_fx2w, the scoring kernel of a GMM system whose models share their variances (the
// reference's UBM + MAP-adapted speaker models, build_spk_models.py:170): replaces the (S+1) runs of
// `gmm-global-get-frame-likes --average=true` of gmm_ubm_kaldiHelper.py:202-221 for every voiced frame of a batch.
// Its own translation unit because it is compiled with -mllvm -amdgpu-mfma-vgpr-form (fakebob_amd/build.py): the
// accumulators live in vector registers, where the logsumexp update reads them without v_accvgpr_read copies, and
// the frame operands that do not fit beside them are parked in the accumulation registers, from where the MFMAs
// read them directly as their B operand at the full rate (tools/probes/mfma_operand_probe.hip).
#include <float.h>
#include <cstdlib>

#include "fb_kernels.h"
#include "gmm_split.h"

// ---------------------------------------------------------------------------------------------
// k_gmm_fx2w: the scoring form of k_gmm_fx2 for ONE variance group (mean-only MAP adaptation: the UBM and all its
// speaker models) -- one wave per SIMD, 64 frames per wave, software-pipelined, the whole tile as straight-line code.
//
// Delta form (round 3).  Means-only MAP adaptation (build_spk_models.py:170, gmm-global-est-map.cc:81) leaves weights
// and variances alone, so for every component k and model m >= 1
//     ll_m,k(x) = ll_0,k(x) + (gconst_m,k - gconst_0,k) + (means_invvars_m,k - means_invvars_0,k) . x
// Items per tile: Q (the shared quadratic term, from zero), model 0 (continues Q's accumulator IN PLACE: afterwards it
// holds model 0's values), then one DELTA item per other model whose MFMAs take model 0's finished accumulator as their
// C operand.  The delta operands are small, so a delta item runs P < 3 of the three partial products of the two-term
// split when fb_load_gmm finds the models close enough (P = 2: both parameter terms against the frames' leading term,
// 20 MFMAs per step instead of 30; the dropped product is 2^-12 |delta . x|): 160 instead of 210 MFMAs per tile for
// UBM + 5 speakers.  The speaker-minus-UBM differences the OSI / SV scores are made of also come out closer to the
// float64 oracle than from independent chains (the base model's rounding is common to both).
//
// What the design rests on (tools/probes/coissue_probe.hip, valu_cost_probe.hip, mfma_operand_probe.hip; one wave per
// SIMD):
//   * MFMAs on ONE accumulator issue only as fast as they execute (the wave sits at the next dependent MFMA), so in
//     k_gmm_fx2 an item's update starts when its 15 MFMAs are done.  MFMAs that alternate between two INDEPENDENT
//     accumulators are queued by the matrix pipe and the wave goes on issuing behind them.
//   * The wave issues in order: a vector instruction overlaps the matrix pipe only if it stands behind an MFMA in the
//     instruction stream.  About five plain VALU instructions per MFMA are free (16.5 ns per MFMA with 0 .. 4
//     v_fma_f32 behind it, 18.5 with 6, 23 with 8); 2 v_fma + 2 v_exp + 2 v_add cost 20 ns: a gap costs
//     max(32 cycles, ~8 for the MFMA's issue + 4 per plain instruction + 8 per v_exp_f32).
//   * A PACKED f32 instruction behind an MFMA stalls the wave: MFMA + one v_pk_fma_f32 = 20.6 ns, the packed form of
//     the update's gap 30.7 ns against 20.3 ns unpacked.  The update is written with single instructions.
//   * An MFMA reads its B operand from an accumulation register as fast as from a vector register, with the C / D
//     operand in either file (13.8 - 14.3 ns per MFMA in all four combinations).
// The two independent chains of an item are the two 32-frame halves of the wave's 64 frames: both use the SAME
// parameter fragments (half the LDS reads per MFMA) and every item, the quadratic one included, is a pair.
// Registers (one wave per SIMD: 512): the accumulators -- the base set and two delta sets, two halves each: 96 -- live
// in VECTOR registers (-amdgpu-mfma-vgpr-form, fakebob_amd/build.py), so the update reads the values in place (the
// 226 v_accvgpr_read copies of the round-2 kernel and the hazard s_nops in front of them are gone); of the 160
// registers of frame operands only the leading term of x (40: every item but Q uses it) stays beside them, x's second
// term and both terms of x^2 (120: used by the two base items only) are parked in accumulation registers by an empty
// asm and read from there by the MFMAs; + 24 of parameter fragments: 201 vector + 120 accumulation registers.
// 4 waves (256 frames) per workgroup, one workgroup per CU, component chunks chosen so that a launch is one round of
// <= 256 workgroups.
// With one wave per SIMD nothing hides instruction fetch after a branch (a loop over items with the item kind,
// pending update and padding decided by branches ran at ~1500 cycles per item with an EMPTY body), hence the
// specialisation: M and P are template parameters, accumulator sets have static roles (delta item m writes set m & 1
// while the previous item's values are updated in the gaps between its MFMAs, fb_fxw_step; the quadratic and the base
// item carry the updates of the previous tile's last two models, the last delta item none), and a tile is one basic block.  Models with C % 32 != 0, several
// variance groups or other M run on k_gmm_fx2.
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
// adds to the step.  Left alone hipcc lumps the ~160 vector instructions of an update behind the MFMAs (kernel time =
// MFMA time + update time, measured), so the step is cut into one scheduling region per MFMA
// (__builtin_amdgcn_sched_barrier(0)) and each region gets its share of the 30 slices of the update of the pending set
// (p0, p1 = the two halves' 16 values each, pm / ps their LDS state [2 halves][256]):
//   slice  0       the state is requested from LDS (the pending values are still in the matrix pipe)
//   slices 2..9    four values are copied from the accumulation registers into vector registers, where they stay for
//                  the second pass (two reads per value would make the update the longer pipe); running maximum
//   slices 10, 11  new reference r = fl(m L), the old sums rescaled
//   slices 12..27  one value of each half: fma, exponential, and the ADD of the previous slice's exponentials (so that
//                  no exponential is consumed right behind itself); even and odd values are summed apart and joined at
//                  the end, which is fb_lse_update16's order
//   slices 28, 29  last adds, state written back
// NP = partial products per K chunk: 3 (a2 b1 + a1 b2 + a1 b1: the full two-term product, 30 MFMAs per step, one slice
// per gap), 2 (a2 b1 + a1 b1: both parameter terms against the leading frame term, 20 MFMAs) or 1 (a1 b1, 10 MFMAs) --
// the delta items of k_gmm_fx2w; with fewer gaps than slices the slices 2 .. 29 are dealt evenly over the gaps 2 ..
// The parameter fragments are streamed with the K chunks -- chunk c + 1 is read from LDS while the MFMAs of chunk
// c run (two alternating register sets for the chunks 1 .. NK-1; chunk 0 has its own, z1 / z2, refilled with the NEXT
// item's chunk 0 when pf0) -- 24 registers instead of two whole items' 80: that is what leaves room for the 32
// pending values.
// The LDS-DMA pieces [dq0, dq0 + dn) of this wave's share of the next parameter group go out one per chunk.
// UPD = false: no pending set (the base model's item: its own values are not there yet).
template <int NK, int NP, bool UPD>
__device__ __forceinline__ void fb_fxw_step(const u32x4 *__restrict__ cur4, const u32x4 *__restrict__ nxt4, const bool pf0,
                                            int lane, u32x4 &z1, u32x4 &z2, const u32x4 (&b1)[2][NK],
                                            const u32x4 (&b2)[2][NK], const f32x16 &init0, const f32x16 &init1,
                                            f32x16 &out0, f32x16 &out1, const f32x16 &p0, const f32x16 &p1,
                                            float *__restrict__ pm, float *__restrict__ ps, float ls,
                                            const u32x4 *__restrict__ dsrc, unsigned ddst, const int dq0, const int dn) {
  constexpr int KP = 2 * NP, NG = KP * NK;  // MFMAs per chunk / per step
  static_assert(NP >= 1 && NP <= 3 && NG >= 10, "slice layout");
  f32x16 x0 = init0, x1 = init1;
  // fragment sets of the chunks 1 .. NK-1, read from LDS PD chunks ahead of their MFMAs: one chunk (6 or 4 MFMAs) covers
  // the LDS latency, but with one product per chunk (2 MFMAs) it takes two; chunk c uses set c % (PD + 1)
  constexpr int PD = NP == 1 ? 2 : 1;
  u32x4 s1[PD + 1], s2[PD + 1];
  float v0[16], v1[16];
  float t0 = FB_GMM_NEG, t1 = FB_GMM_NEG, mo0 = 0.f, mo1 = 0.f, so0 = 0.f, so1 = 0.f, mn0 = 0.f, mn1 = 0.f;
  float nr0 = 0.f, nr1 = 0.f, d0 = 0.f, d1 = 0.f;
  float se0 = 0.f, se1 = 0.f, sd0 = 0.f, sd1 = 0.f;  // sums of the even / odd values, halves 0 / 1
  float e0 = 0.f, e1 = 0.f;                          // the previous slice's exponentials
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c = g / KP, kk = g % KP;
    const int k = NP == 3 ? kk : (NP == 2 ? (kk < 2 ? kk : kk + 2) : kk + 4);  // which of the six products of a chunk
    if (kk == 0) {
#pragma unroll
      for (int cn = (c == 0 ? 1 : c + PD); cn <= c + PD && cn < NK; ++cn) {  // (the step's first gap requests 1 .. PD)
        s1[cn % (PD + 1)] = cur4[(0 * NK + cn) * 64 + lane];
        if (NP >= 2) s2[cn % (PD + 1)] = cur4[(1 * NK + cn) * 64 + lane];
      }
    }
    const u32x4 &a1 = c == 0 ? z1 : s1[c % (PD + 1)], &a2 = c == 0 ? z2 : s2[c % (PD + 1)];
    if (k == 0) FB_FX_MFMA(a2, b1[0][c], x0);
    else if (k == 1) FB_FX_MFMA(a2, b1[1][c], x1);
    else if (k == 2) FB_FX_MFMA(a1, b2[0][c], x0);
    else if (k == 3) FB_FX_MFMA(a1, b2[1][c], x1);
    else if (k == 4) FB_FX_MFMA(a1, b1[0][c], x0);
    else FB_FX_MFMA(a1, b1[1][c], x1);
    if (pf0 && g == KP + 1) {  // chunk 0's registers are free: the next item's chunk 0 (pf0 is a constant after unrolling)
      z1 = nxt4[(0 * NK + 0) * 64 + lane];
      z2 = nxt4[(1 * NK + 0) * 64 + lane];
    }
    if (kk == (KP > 3 ? 3 : KP - 1) && c < dn) fb_glds16(dsrc + (dq0 + c) * 64, ddst + (unsigned)(dq0 + c) * 1024u);
    if constexpr (UPD) {
      // slices of this gap: 0 and 1 in the gaps 0 and 1, the 28 others dealt over the NG - 2 gaps that follow
      const int sl0 = g < 2 ? g : 2 + ((g - 2) * 28) / (NG - 2), sl1 = g < 2 ? g + 1 : 2 + ((g - 1) * 28) / (NG - 2);
#pragma unroll
      for (int sl = sl0; sl < sl1; ++sl) {
        if (sl == 0) { mo0 = pm[0]; mo1 = pm[256]; so0 = ps[0]; so1 = ps[256]; }
        if (sl >= 2 && sl < 10) {
          const int r = 2 * (sl - 2);
          v0[r] = p0[r]; v0[r + 1] = p0[r + 1]; v1[r] = p1[r]; v1[r + 1] = p1[r + 1];
          // the copies are the compiler's (it knows the matrix pipe's hazards); the empty asm keeps them HERE and in
          // vector registers
          asm volatile("" : "+v"(v0[r]), "+v"(v0[r + 1]), "+v"(v1[r]), "+v"(v1[r + 1]));
          t0 = fb_v_max3(t0, v0[r], v0[r + 1]);
          t1 = fb_v_max3(t1, v1[r], v1[r + 1]);
        } else if (sl == 10) {
          mn0 = fb_v_max3(mo0, t0, t0); mn1 = fb_v_max3(mo1, t1, t1);
          const float rn0 = fb_v_mul(mn0, ls), rn1 = fb_v_mul(mn1, ls);
          nr0 = -rn0; nr1 = -rn1;
          d0 = fb_v_fma(mo0, ls, nr0); d1 = fb_v_fma(mo1, ls, nr1);  // r_old - r_new; r_old = -inf at the start
        } else if (sl == 11) {
          e0 = fb_v_exp(d0); e1 = fb_v_exp(d1);
        } else if (sl >= 12 && sl < 28) {
          const int r = sl - 12;
          const float u0 = fb_v_fma(v0[r], ls, nr0), u1 = fb_v_fma(v1[r], ls, nr1);
          const float f0 = fb_v_exp(u0), f1 = fb_v_exp(u1);
          if (r == 0) { se0 = fb_v_mul(so0, e0); se1 = fb_v_mul(so1, e1); }      // s_old * 2^(r_old - r_new)
          else if (r == 1) { se0 = fb_v_add(se0, e0); se1 = fb_v_add(se1, e1); }  // + value 0
          else if (r == 2) { sd0 = e0; sd1 = e1; }                                 // value 1 starts the odd sums
          else if (r & 1) { se0 = fb_v_add(se0, e0); se1 = fb_v_add(se1, e1); }  // value r - 1 is even
          else { sd0 = fb_v_add(sd0, e0); sd1 = fb_v_add(sd1, e1); }
          e0 = f0; e1 = f1;
        } else if (sl == 28) {
          sd0 = fb_v_add(sd0, e0); sd1 = fb_v_add(sd1, e1);                       // value 15
        } else if (sl == 29) {
          pm[0] = mn0; pm[256] = mn1;
          ps[0] = fb_v_add(se0, sd0); ps[256] = fb_v_add(se1, sd1);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("" : "+v"(x0), "+v"(x1));  // the accumulators stay in vector registers (see the file header)
  out0 = x0;
  out1 = x1;
}

template <int NK, int M, int P>
__global__ __launch_bounds__(256, 1) void k_gmm_fx2w(FbGmmDev g, const float *__restrict__ feats,
                                                     const int *__restrict__ n_rows_ptr, int tiles_per_chunk,
                                                     int rows_cap, float *__restrict__ part_m,
                                                     float *__restrict__ part_s, int xcd_map) {
  if (g.stop && *g.stop) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG4 = 2 * NK * 64;  // 16-byte units per item
  constexpr int NI = M + 1;          // items per tile: Q, base model, delta_1 .. delta_{M-1}
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
        // only the base items use these: parked in accumulation registers, read from there by their MFMAs
        asm volatile("" : "+a"(bx2[hf][c]), "+a"(bq1[hf][c]), "+a"(bq2[hf][c]));
      }
  }
#pragma unroll
  for (int m = 0; m < 2 * M; ++m) { st_m[m * 256 + tid] = FB_GMM_NEG; st_s[m * 256 + tid] = 0.0f; }

  const int tile0 = chunk_i * tiles_per_chunk;
  const int tile1 = min(g.n_tiles, tile0 + tiles_per_chunk);
  const int n_t = tile1 - tile0, total_items = n_t * NI;
  const u32x4 *gimg = g.images_fd + (size_t)tile0 * NI * IMG4;
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
    // The base steps of the FIRST tile have no finished models to carry, and a branch for them costs more than a bogus
    // update (nothing hides an instruction fetch at one wave per SIMD): they "update" the last two models with 16
    // sentinel values -2^60 / unscale, whose scaled form is exactly -2^60 fl(log2 e) -- the state becomes (that
    // maximum, 16), and the first real update rescales those 16 by 2^(-1.6e18) = 0.
    const float sentinel = -fb_pow2f(60 - sh + g.kacc);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = sentinel; acc[0][1][r] = sentinel; acc[1][0][r] = sentinel; acc[1][1][r] = sentinel; }
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
#pragma clang loop unroll(full)
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
      //   jj = 0      quadratic item -> hq (from zero); carries the update of the previous tile's last model
      //   jj = 1      base model: hq continues IN PLACE, after the step it holds the base model's values; no update
      //   jj = 1 + m  delta item of model m >= 1: P products per K chunk on top of the base model's values (hq is
      //               the MFMA's C operand, the accumulator set m & 1 its destination); carries the update of model
      //               m - 1 -- for m = 1 that is the base model, read from hq itself
      const bool pf0 = (jj + 1 < NI && jj + 1 != GA);
      const u32x4 *nxt4 = slot0 + item4(jj + 1 < NI ? jj + 1 : 0);
      // Which step carries which update (DEFER: three or more models).  The base items have the vector issue port to
      // spare (30 MFMAs each), the delta items do not (10 - 30 MFMAs for the 126 instructions of an update), so the
      // updates of the LAST TWO models wait for the next tile's Q and base steps -- their accumulator sets are not written
      // again before delta items 1 / 2 of that tile -- and the last delta item carries none.
      constexpr bool DEFER = M >= 3;
      if (jj == 0) {
        if constexpr (DEFER) {
          float *pm = st_m + (2 * (M - 2)) * 256 + tid, *ps = st_s + (2 * (M - 2)) * 256 + tid;
          fb_fxw_step<NK, 3, true>(cur4, nxt4, pf0, lane, z1, z2, bq1, bq2, zero, zero, hq[0], hq[1], acc[(M - 2) & 1][0], acc[(M - 2) & 1][1], pm, ps, ls, dsrc, ddst, dq0, dn);
        } else {
          fb_fxw_step<NK, 3, false>(cur4, nxt4, pf0, lane, z1, z2, bq1, bq2, zero, zero, hq[0], hq[1], zero, zero, st_m, st_s, ls, dsrc, ddst, dq0, dn);
        }
      } else if (jj == 1) {
        float *pm = st_m + (2 * (M - 1)) * 256 + tid, *ps = st_s + (2 * (M - 1)) * 256 + tid;
        fb_fxw_step<NK, 3, true>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], hq[0], hq[1], acc[(M - 1) & 1][0], acc[(M - 1) & 1][1], pm, ps, ls, dsrc, ddst, dq0, dn);
      } else if (jj == 2) {
        float *pm = st_m + tid, *ps = st_s + tid;
        fb_fxw_step<NK, P, true>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[1][0], acc[1][1], hq[0], hq[1], pm, ps, ls,
                                 dsrc, ddst, dq0, dn);
      } else if (DEFER && jj == NI - 1) {
        fb_fxw_step<NK, P, false>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[(jj - 1) & 1][0], acc[(jj - 1) & 1][1],
                                  zero, zero, st_m, st_s, ls, dsrc, ddst, dq0, dn);
      } else {
        float *pm = st_m + (2 * (jj - 2)) * 256 + tid, *ps = st_s + (2 * (jj - 2)) * 256 + tid;
        fb_fxw_step<NK, P, true>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[(jj - 1) & 1][0], acc[(jj - 1) & 1][1],
                                 acc[(jj - 2) & 1][0], acc[(jj - 2) & 1][1], pm, ps, ls, dsrc, ddst, dq0, dn);
      }
      if (jj == GA - 1 || jj == NI - 1) publish();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (M >= 3) update(acc[(M - 2) & 1][0], acc[(M - 2) & 1][1], M - 2);
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

template <int NK, int M, int P>
static void launch_gmm_fxw_p(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
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
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gmm_fx2w<NK, M, P>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) == hipSuccess)
      optin.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_gmm_fx2w<NK, M, P>), grid, dim3(256), ldsb, s, g, feats, n_rows_ptr, tpc, rows_cap, part_m, part_s,
                     xcd_map);
}
template <int NK, int M>
static void launch_gmm_fxw_t(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                             int rows_cap, int n_chunks, int tpc, float *part_m, float *part_s) {
  switch (g.delta_p) {  // products per K chunk of the delta items, chosen by fb_load_gmm
    case 1: launch_gmm_fxw_p<NK, M, 1>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 2: launch_gmm_fxw_p<NK, M, 2>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    default: launch_gmm_fxw_p<NK, M, 3>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
  }
}
// k_gmm_fx2w is instantiated for the shapes the reference's systems have with the recipe's 72-dimensional features
// (NKF = 5): one variance group, every component tile full, 2 <= M <= FB_FXW_MAX_M models (SV: UBM + 1; OSI: UBM + up
// to 9 speakers; CSI: up to 10 speakers).  Everything else runs on k_gmm_fx2.
#define FB_FXW_MAX_M 10  // (1 + M) 10 KB items + the state of 2 M x 256 frames: 156 KB of LDS at M = 10
bool fb_gmm_use_wide(const FbGmmDev &g) {
  const bool off = getenv("FB_GMM_NARROW") != nullptr;  // read per call: the tests switch it inside one process
  return g.mode == FB_GMM_MODE_FX2 && !off && g.NKF == 5 && g.n_items == g.M + 1 && (g.C & 31) == 0 && g.M >= 2 &&
         g.M <= FB_FXW_MAX_M && g.item_model_host_q_first && g.delta_p >= 1 && g.images_fd != nullptr;
}
void fb_launch_gmm_wide(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap,
                        int n_chunks, int tpc, float *part_m, float *part_s) {
  switch (g.M) {
    case 2: launch_gmm_fxw_t<5, 2>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 3: launch_gmm_fxw_t<5, 3>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 4: launch_gmm_fxw_t<5, 4>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 5: launch_gmm_fxw_t<5, 5>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 6: launch_gmm_fxw_t<5, 6>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 7: launch_gmm_fxw_t<5, 7>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 8: launch_gmm_fxw_t<5, 8>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 9: launch_gmm_fxw_t<5, 9>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    case 10: launch_gmm_fxw_t<5, 10>(s, g, feats, n_rows_ptr, rows_cap, n_chunks, tpc, part_m, part_s); break;
    default: break;  // fb_gmm_use_wide() admits only the cases above
  }
}

