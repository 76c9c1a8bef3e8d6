// gmm_wide_kernel.hip -- k_gmm_fx2w, the scoring kernel of a GMM system whose models share their variances (the
// reference's UBM + MAP-adapted speaker models, build_spk_models.py:170): replaces the (S+1) runs of
// `gmm-global-get-frame-likes --average=true` of gmm_ubm_kaldiHelper.py:202-221 for every voiced frame of a batch.
// Its own translation unit because it is compiled with -mllvm -amdgpu-mfma-vgpr-form (fakebob_amd/build.py): the
// accumulators live in vector registers, where the logsumexp update reads them without v_accvgpr_read copies, and
// the frame operands that do not fit beside them are parked in the accumulation registers, from where the MFMAs
// read them directly as their B operand at the full rate (tools/probes/mfma_operand_probe.hip).
#include <float.h>
#include <cstdlib>
#include <type_traits>

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
// Log2-domain accumulators (round 3, second step).  The images are the parameters times log2 e, balanced per dimension
// against the frames by exact powers of two instead of scaled as a whole (fb_load_gmm), and the frame operand carries
// -R, an integer reference per frame, in two K places of the padding: an accumulator is t = ll log2 e - R, ready for
// v_exp_f32.  The logsumexp update is one exponential and one addition per value (above fb_fxw_step), with a cold
// rescue path for the frames whose reference turns out too low and for frames outside f16's range.
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
// asm and read from there by the MFMAs; + 24 of parameter fragments: 204 vector + 120 accumulation registers.
// 4 waves (256 frames) per workgroup, one workgroup per CU, component chunks chosen so that a launch is one round of
// <= 256 workgroups.
// With one wave per SIMD nothing hides instruction fetch after a branch (a loop over items with the item kind,
// pending update and padding decided by branches ran at ~1500 cycles per item with an EMPTY body), hence the
// specialisation: M and P are template parameters, accumulator sets have static roles (delta item m writes set m & 1
// while the previous item's values are updated in the gaps between its MFMAs, fb_fxw_step; the quadratic and the base
// item carry the updates of the previous tile's last two models, the last delta item none), and a tile is a straight
// line of basic blocks whose only branches are the never-taken ones to the rescue code behind the loop.  Models with
// C % 32 != 0, D % 4 != 0, several variance groups or other M run on k_gmm_fx2.
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

// The logsumexp state of k_gmm_fx2w (round 3, second form).  The parameter images carry log2 e (fb_load_gmm) and the
// quadratic item's accumulation includes -R, R an integer reference per frame that stands in two K places of the x^2
// operand's padding (set_ref in the kernel): a finished accumulator holds t = ll * log2 e - R, and a value costs ONE
// v_exp_f32 and ONE v_add_f32 -- no running maximum, no subtraction (204 v_fma_f32 and 108 v_max3_f32 of the 850
// vector instructions per tile gone; the steps of the delta items are bound by the vector issue port).
//   state per (frame half, model) in LDS: (mref, s), sum_k 2^(ll_k log2 e) = s * 2^mref; mref an integer-valued float,
//                 equal to the frame's R wherever a fast update meets it (the tile boundary that moves R moves every
//                 model's state along; the fast path never reads mref)
//   fast update:  s += sum_i 2^t_i        [times 2^(R_old - R_new) for the two updates deferred across a moved reference]
//   guard:        s <= 2^100 (false for inf and NaN too), else the RESCUE below redoes the update from the accumulator
//                 registers -- still there -- the classical way (maximum, re-reference) and proposes a new R for the
//                 next tile.  Cold code behind the loop (~2.4 us per visit, mostly instruction fetch), taken by the
//                 whole wave when one lane needs it.
// R starts from a guaranteed LOWER bound of every model's log2 sum: the best of the log-likelihoods of
// FB_FXW_ANCHORS wide components of the base model (the "anchors", fb_load_gmm), evaluated in the prologue on the
// frame operand's leading term, minus a Cauchy-Schwarz bound of what the other models' deltas can take away, +
// FB_FXW_ROFF: never more than 2^64 above a sum (nothing that matters underflows: what is flushed lies 2^-62 below
// the ANCHOR's term; Kaldi's own LogSumExp drops what is 2^-23 below the maximum), and a sum may lie 2^164 = 113 nats
// above the bound before a rescue is needed.  On the SURVEY.md 8(d) workload: 56 rescues in 87 552 updates with two
// anchors (3 563 with one -- 9 us of the launch).
// A wave with a frame that needed the dynamic range shift (|x| >= 181 spreads) takes the rescue path for every update.
#ifdef FB_FXW_COUNT  // instrumented build (tools/profile/fxw_instrumented.sh): how often the cold paths run
__device__ unsigned long long g_fxw_counts[4];
extern "C" int fb_debug_fxw_counts(unsigned long long *out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fxw_counts), sizeof(g_fxw_counts)) != hipSuccess) return -1;
  unsigned long long z[4] = {0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fxw_counts), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#define FXW_COUNT(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_fxw_counts[i], 1ull); } while (0)
#else
#define FXW_COUNT(i) do { } while (0)
#endif
#ifdef FB_FXW_STAMP  // instrumented build (tools/profile/fxw_instrumented.sh): where a workgroup's time goes
__device__ unsigned long long g_fxw_stamps[16];
extern "C" int fb_debug_fxw_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fxw_stamps), sizeof(g_fxw_stamps)) == hipSuccess ? 0 : -1;
}
#define FXW_STAMP(i) do { if (blockIdx.x == 8 && threadIdx.x == 0) g_fxw_stamps[i] = wall_clock64(); } while (0)
#else
#define FXW_STAMP(i) do { } while (0)
#endif
#define FB_FXW_ROFF 64.0f
#define FB_FXW_SMAX 1.2676506002282294e30f  // 2^100
struct FbFxwUpd { float so0, so1, sn0, sn1; };  // sums before / after a fast update (halves 0, 1)

// the rescue: fold the 16 values p (t * 2^-sh, relative to rt) into (mo, so) with a maximum; lanes with `need` store
__device__ __forceinline__ void fb_fxw_slow_update(const f32x16 &p, float up, float rt, float mo, float so,
                                                   float *__restrict__ pm, float *__restrict__ ps, float &rn, bool need) {
  float tm = -3.0e38f;  // (finite also for the sentinel values of the first tile)
#pragma unroll
  for (int r = 0; r < 16; ++r) tm = fmaxf(tm, __fmul_rn(p[r], up));
  const float mo_e = so > 0.0f ? mo : -INFINITY;              // an empty state has no reference yet
  const float mn = fmaxf(mo_e, __fadd_rn(rt, rintf(tm)));     // integer-valued like mo and rt
  const float off = __fsub_rn(mn, rt);
  float a0 = __fmul_rn(so, __builtin_amdgcn_exp2f(__fsub_rn(mo_e, mn))), a1 = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    a0 = __fadd_rn(a0, __builtin_amdgcn_exp2f(__fmaf_rn(p[r], up, -off)));
    a1 = __fadd_rn(a1, __builtin_amdgcn_exp2f(__fmaf_rn(p[r + 1], up, -off)));
  }
  if (need) {
    *pm = mn;
    *ps = __fadd_rn(a0, a1);
    rn = fmaxf(rn, mn);
  }
}

// One item step with the logsumexp update of ANOTHER accumulator set threaded between the MFMAs, by construction.
// One wave per SIMD issues in order; about five plain vector instructions behind an MFMA are free, what exceeds them
// adds to the step.  Left alone hipcc lumps the vector instructions of an update behind the MFMAs (kernel time =
// MFMA time + update time, measured), so the step is cut into one scheduling region per MFMA
// (__builtin_amdgcn_sched_barrier(0)) and each region gets its share of the 20 slices of the update of the pending set
// (p0, p1 = the two halves' 16 values each, ps their sums in LDS [2 halves][256]; DEFERRED: the set belongs to the
// previous tile, wd0 / wd1 = 2^(that tile's reference - this one's), 1 unless a rescue came between):
//   slice  0       the sums are requested from LDS
//   slices 1..16   one value of each half: its exponential
//   slices 3..18   the ADD of the exponentials issued two slices earlier -- in another gap, so that no addition reads
//                  a transcendental result with nothing but transcendentals in between (an s_nop each); even and odd
//                  values are summed apart
//   slice  19      sums written back; the caller checks them (guard above)
// NP = partial products per K chunk: 3 (a2 b1 + a1 b2 + a1 b1: the full two-term product, 30 MFMAs per step), 2
// (a2 b1 + a1 b1: both parameter terms against the leading frame term, 20 MFMAs) or 1 (a1 b1, 10 MFMAs) -- the delta
// items of k_gmm_fx2w.
// The parameter fragments are streamed with the K chunks -- chunk c + 1 is read from LDS while the MFMAs of chunk
// c run (two alternating register sets for the chunks 1 .. NK-1; chunk 0 has its own, z1 / z2, refilled with the NEXT
// item's chunk 0 when pf0) -- 24 registers instead of two whole items' 80.
// The LDS-DMA pieces [dq0, dq0 + dn) of this wave's share of the next parameter group go out one per chunk.
// UPD = false: no pending set.
template <int NK, int NP, bool UPD, bool ZI = false, bool DEFERRED = false>
__device__ __forceinline__ void fb_fxw_step(const u32x4 *__restrict__ cur4, const u32x4 *__restrict__ nxt4, const bool pf0,
                                            int lane, u32x4 &z1, u32x4 &z2, const u32x4 (&b1)[2][NK],
                                            const u32x4 (&b2)[2][NK], const f32x16 &init0, const f32x16 &init1,
                                            f32x16 &out0, f32x16 &out1, const f32x16 &p0, const f32x16 &p1,
                                            float *__restrict__ ps, const float wd0, const float wd1, FbFxwUpd &u,
                                            const u32x4 *__restrict__ dsrc, unsigned ddst, const int dq0, const int dn) {
  constexpr int KP = 2 * NP, NG = KP * NK;  // MFMAs per chunk / per step
  constexpr int NS = 20;                    // slices of an update
  static_assert(NP >= 1 && NP <= 3 && NG >= 10, "slice layout");
  f32x16 x0, x1;  // ZI: from zero -- a literal C operand of the first MFMAs, no registers to clear
  if constexpr (ZI) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { x0[r] = 0.0f; x1[r] = 0.0f; }
  } else {
    x0 = init0;
    x1 = init1;
  }
  // fragment sets of the chunks 1 .. NK-1, read from LDS PD chunks ahead of their MFMAs: one chunk (6 or 4 MFMAs) covers
  // the LDS latency, but with one product per chunk (2 MFMAs) it takes two; chunk c uses set c % (PD + 1)
  constexpr int PD = NP == 1 ? 2 : 1;
  u32x4 s1[PD + 1], s2[PD + 1];
  float se0 = 0.f, se1 = 0.f, sd0 = 0.f, sd1 = 0.f;  // sums of the even / odd values, halves 0 / 1
  float ea0 = 0.f, ea1 = 0.f, eb0 = 0.f, eb1 = 0.f;  // the exponentials of the last two slices (a: older)
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
      // the 20 slices as 80 single instructions -- a slice = {addition half 0, addition half 1, exponential half 0,
      // exponential half 1} --, dealt evenly over the gaps: with 30 MFMAs per step a gap carries two or three of them
      // (whole slices put four into two gaps of three and stretched each by ~3 ns -- valu_cost_probe's "2 v_exp + 2 v_add
      // behind an MFMA" --: k_gmm_fx2w with three products per delta item 99 -> 82 us, round 4)
      const int m0 = (g * 4 * NS) / NG, m1 = ((g + 1) * 4 * NS) / NG;
#pragma unroll
      for (int mm = m0; mm < m1; ++mm) {
        const int sl = mm >> 2, part = mm & 3;
        // value r's exponential is added TWO slices after it was issued: an addition must not read a transcendental
        // result with only transcendentals in between -- an s_nop each, 70 per tile
        const int r = sl - 1;          // the value whose exponential this slice issues (0 .. 15)
        const int ra = sl - 3;         // the value this slice adds (its exponential is ea)
        if (part == 0) {
          if (ra == 0) se0 = ea0;                                        // value 0 starts the even sums
          else if (ra == 1) sd0 = ea0;                                   // value 1 the odd ones
          else if (ra >= 2 && ra <= 15 && !(ra & 1)) se0 = fb_v_add(se0, ea0);
          else if (ra >= 2 && ra <= 15) sd0 = fb_v_add(sd0, ea0);
        } else if (part == 1) {
          if (ra == 0) se1 = ea1;
          else if (ra == 1) sd1 = ea1;
          else if (ra >= 2 && ra <= 15 && !(ra & 1)) se1 = fb_v_add(se1, ea1);
          else if (ra >= 2 && ra <= 15) sd1 = fb_v_add(sd1, ea1);
          ea0 = eb0; ea1 = eb1;
        } else if (part == 2) {
          if (sl == 0) { u.so0 = ps[0]; u.so1 = ps[256]; }
          if (r >= 0 && r <= 15) eb0 = fb_v_exp(p0[r]);
        } else {
          if (r >= 0 && r <= 15) eb1 = fb_v_exp(p1[r]);
          if (sl == NS - 1) {  // (value 15 was added in this slice's first two parts)
            // the state is relative to the set's reference unless a rescue moved the frame's reference between the tile
            // the set belongs to and this one: wd is 2^(old - new) then, 1 otherwise (kernel, tile boundary)
            u.sn0 = DEFERRED ? fb_v_fma(fb_v_add(se0, sd0), wd0, u.so0) : fb_v_add(fb_v_add(se0, sd0), u.so0);
            u.sn1 = DEFERRED ? fb_v_fma(fb_v_add(se1, sd1), wd1, u.so1) : fb_v_add(fb_v_add(se1, sd1), u.so1);
            ps[0] = u.sn0; ps[256] = u.sn1;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("" : "+v"(x0), "+v"(x1));  // the accumulators stay in vector registers (see the file header)
  out0 = x0;
  out1 = x1;
}


// The F6 form of a delta step (round 4): the leading f16 product (2 NK MFMAs, as NP = 1) and then the two products NP = 1
// leaves out -- delta_2 . x_1 and delta_1 . x_2 -- as four v_mfma_scale_f32_32x32x64_f8f6f4 per half on block-scaled
// fp6 / fp4 operands: K = 64 each at the price of one K = 16 f16 MFMA (tools/probes/f6_mfma_probe.hip,
// f4f6_mfma_probe.hip: ~20 against 22.5 ns), 18 MFMAs per step against NP = 3's 30.  The frames' side is e2m3 (4
// significant bits: its rounding differs from frame to frame and averages out over an utterance); delta_2, whose
// rounding is the same for every frame, gets TWO terms (e2m3 + e2m1 of what that leaves), delta_1 -- against the
// zero-mean x_2 -- one.  Operands (fb_load_gmm's F6 item layout): a lane holds the K places 8 h .. 8 h + 7 of every chunk,
// as in the f16 fragments, so the frames' side is made from the f16 fragments the lane already has (the kernel's
// prologue, v_cvt_scalef32_pk32_fp6_f16) and the parameters' side comes from the item's second half in LDS.  MFMAs of a
// half: block 0 (delta_2 hi) x fq[0] (x_1), block L (delta_2 lo, fp4) x fq[0], block 1 (delta_1 2^-12) x fq[1] (x_2 2^12),
// block 2 (the dimensions of chunk 4: lanes h = 0, the frames' operand is zero in the others) x fq[2].  The update slices
// of the pending set are dealt over the 18 gaps as in fb_fxw_step.
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int NK, bool UPD>
__device__ __forceinline__ void fb_fxw_step6(const u32x4 *__restrict__ cur4, const u32x4 *__restrict__ nxt4, const bool pf0,
                                             int lane, u32x4 &z1, const u32x4 (&b1)[2][NK], const i32x8 (&fq)[2][3],
                                             const int (&fsc)[2][3], const f32x16 &init0, const f32x16 &init1,
                                             f32x16 &out0, f32x16 &out1, const f32x16 &p0, const f32x16 &p1,
                                             float *__restrict__ ps, FbFxwUpd &u, const u32x4 *__restrict__ dsrc, unsigned ddst,
                                             const int dq0, const int dn) {
  static_assert(NK == 5, "F6 item layout");
  constexpr int NM = 2 * NK, NG = NM + 8;  // f16 MFMAs / MFMAs per step
  constexpr int NS = 20, PD = 2;
  f32x16 x0 = init0, x1 = init1;
  u32x4 s1[PD + 1];
  i32x8 pa[4];  // blocks 0, L, 1, 2 in the order of their MFMAs
  int psc = 0, psh[4] = {0, 0, 0, 0};  // the lane's four scale bytes; each one in byte 0 (what the instruction takes)
  float se0 = 0.f, se1 = 0.f, sd0 = 0.f, sd1 = 0.f;
  float ea0 = 0.f, ea1 = 0.f, eb0 = 0.f, eb1 = 0.f;
  const unsigned char *ib = reinterpret_cast<const unsigned char *>(cur4);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g < NM) {
      const int c = g / 2, kk = g % 2;
      if (kk == 0) {
#pragma unroll
        for (int cn = (c == 0 ? 1 : c + PD); cn <= c + PD && cn < NK; ++cn) s1[cn % (PD + 1)] = cur4[cn * 64 + lane];
        if (c == 0) {
          const u32x4 lo = *reinterpret_cast<const u32x4 *>(ib + 5120 + lane * 16);
          const uint2 hi = *reinterpret_cast<const uint2 *>(ib + 6144 + lane * 8);
          pa[0] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi.x, (int)hi.y, 0, 0};
          psc = *reinterpret_cast<const int *>(ib + 9984 + lane * 4);
        } else if (c == 1) {
          const u32x4 lo = *reinterpret_cast<const u32x4 *>(ib + 8192 + lane * 16);
          pa[1] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], 0, 0, 0, 0};
        } else if (c == 2) {
          const u32x4 lo = *reinterpret_cast<const u32x4 *>(ib + 6656 + lane * 16);
          const uint2 hi = *reinterpret_cast<const uint2 *>(ib + 7680 + lane * 8);
          pa[2] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi.x, (int)hi.y, 0, 0};
        } else if (c == 3) {  // (lanes h = 1 read their component's h = 0 operand: it meets zeros)
          const u32x4 lo = *reinterpret_cast<const u32x4 *>(ib + 9216 + (lane & 31) * 16);
          const uint2 hi = *reinterpret_cast<const uint2 *>(ib + 9728 + (lane & 31) * 8);
          pa[3] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi.x, (int)hi.y, 0, 0};
        } else {  // (well ahead of the MFMAs that read them: no hazard wait states)
          psh[0] = psc; psh[1] = psc >> 16; psh[2] = psc >> 8; psh[3] = psc >> 24;
        }
      }
      const u32x4 &a1 = c == 0 ? z1 : s1[c % (PD + 1)];
      if (kk == 0) FB_FX_MFMA(a1, b1[0][c], x0);
      else FB_FX_MFMA(a1, b1[1][c], x1);
      if (pf0 && g == 3) z1 = nxt4[lane];  // chunk 0's register is free: the next item's chunk 0 (an F6 item again)
      if (kk == 1 && c < dn) fb_glds16(dsrc + (dq0 + c) * 64, ddst + (unsigned)(dq0 + c) * 1024u);
    } else {
      const int blk = (g - NM) / 2, hf = (g - NM) % 2;  // 0: block 0, 1: block L, 2: block 1, 3: block 2
      const int fb = blk == 0 || blk == 1 ? 0 : blk - 1;  // the frames' operand it meets
      const int sa = psh[blk];
      if (blk == 1) {
        if (hf == 0) x0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa[1], fq[0][fb], x0, 4, 2, 0, sa, 0, fsc[0][fb]);
        else x1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa[1], fq[1][fb], x1, 4, 2, 0, sa, 0, fsc[1][fb]);
      } else {
        if (hf == 0) x0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa[blk], fq[0][fb], x0, 2, 2, 0, sa, 0, fsc[0][fb]);
        else x1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa[blk], fq[1][fb], x1, 2, 2, 0, sa, 0, fsc[1][fb]);
      }
    }
    if constexpr (UPD) {  // the slices of fb_fxw_step, four or five instructions per gap
      const int m0 = (g * 4 * NS) / NG, m1 = ((g + 1) * 4 * NS) / NG;
#pragma unroll
      for (int mm = m0; mm < m1; ++mm) {
        const int sl = mm >> 2, part = mm & 3;
        const int r = sl - 1, ra = sl - 3;
        if (part == 0) {
          if (ra == 0) se0 = ea0;
          else if (ra == 1) sd0 = ea0;
          else if (ra >= 2 && ra <= 15 && !(ra & 1)) se0 = fb_v_add(se0, ea0);
          else if (ra >= 2 && ra <= 15) sd0 = fb_v_add(sd0, ea0);
        } else if (part == 1) {
          if (ra == 0) se1 = ea1;
          else if (ra == 1) sd1 = ea1;
          else if (ra >= 2 && ra <= 15 && !(ra & 1)) se1 = fb_v_add(se1, ea1);
          else if (ra >= 2 && ra <= 15) sd1 = fb_v_add(sd1, ea1);
          ea0 = eb0; ea1 = eb1;
        } else if (part == 2) {
          if (sl == 0) { u.so0 = ps[0]; u.so1 = ps[256]; }
          if (r >= 0 && r <= 15) eb0 = fb_v_exp(p0[r]);
        } else {
          if (r >= 0 && r <= 15) eb1 = fb_v_exp(p1[r]);
          if (sl == NS - 1) {
            u.sn0 = fb_v_add(fb_v_add(se0, sd0), u.so0);
            u.sn1 = fb_v_add(fb_v_add(se1, sd1), u.so1);
            ps[0] = u.sn0; ps[256] = u.sn1;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("" : "+v"(x0), "+v"(x1));
  out0 = x0;
  out1 = x1;
}

template <int NK, int M>
__global__ __launch_bounds__(256, 1) void k_gmm_fx2w(FbGmmDev g, const float *__restrict__ feats,
                                                     const int *__restrict__ n_rows_ptr, int n_chunks,
                                                     int rows_cap, float *__restrict__ part_m,
                                                     float *__restrict__ part_s, int xcd_map) {
  if (g.stop && *g.stop) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG4 = 2 * NK * 64;  // 16-byte units per item
  constexpr int NI = M + 1;          // items per tile: Q, base model, delta_1 .. delta_{M-1}
  constexpr int GA = (NI + 1) / 2, GB = NI - GA;  // a tile's items live in two LDS slots: A = items 0 .. GA-1, B = the rest
  FXW_STAMP(0);
  const int n_rows = *n_rows_ptr;
  // A workgroup scores g.fxw_sub (1 or 2) of the n_chunks component chunks of its strip, one after the other: gchunk,
  // gchunk + n_chunks / fxw_sub -- each exactly as a workgroup of its own would (same tiles in the same order from the same
  // fresh state: the same partial sums, bit for bit), behind ONE prologue.  fxw_sub = 2 is the launch of a GPU shared by three or
  // more attacks: half as many workgroups as compute units (fb_engine.hip, run_scoring).
  int strip_i, gchunk;  // XCD-aware (strip, chunk) mapping, as in k_gmm_bx3
  if (xcd_map) {
    const int lin = blockIdx.x, per = 8 / xcd_map;
    const int xcd = lin & 7, idx = lin >> 3;
    gchunk = xcd / per;
    strip_i = idx * per + (xcd % per);
  } else {
    strip_i = blockIdx.x;
    gchunk = blockIdx.y;
  }
  const int n_sub = g.fxw_sub > 1 ? g.fxw_sub : 1, grid_chunks = n_chunks / n_sub;
  const int strip0 = strip_i * 256;
  if (strip0 >= n_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  constexpr int PAD4 = 192;                    // 3 KB behind each slot: fb_fxw_fetch rounds a group up to whole waves
  constexpr int SLOTB4 = GA * IMG4 + PAD4;     // slot A = items 0 .. GA-1 at 0, slot B = the rest
  const u32x4 *slot0 = reinterpret_cast<const u32x4 *>(lds);
  float *st_m = lds + (NI * IMG4 + 2 * PAD4) * 4;  // [M][2 halves][256]
  float *st_s = st_m + M * 512;                // [M][2 halves][256]

  // ---- frame fragments of the two 32-frame halves.  All of a lane's feature loads go out first; while they are under
  //      way the per-dimension tables (balancing factors, anchor components: fb_load_gmm) are staged in LDS -- in slot B,
  //      which no LDS-DMA touches before the first publish() below, a barrier every wave passes after its last read here.
  u32x4 bx1[2][NK], bx2[2][NK], bq1[2][NK], bq2[2][NK];
  int rows[2];
  float4 ft[NK][2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    rows[hf] = strip0 + w * 64 + hf * 32 + j;
    const float *fr = feats + (size_t)(rows[hf] < n_rows ? rows[hf] : 0) * g.D;
#pragma unroll
    for (int c = 0; c < NK; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u) ft[c][hf][u] = *reinterpret_cast<const float4 *>(fr + min(16 * c + 8 * h + 4 * u, g.D - 4));
  }
  // tables: [0, 16 NK) 2^kd, [16 NK, 32 NK) 2^kq, then per anchor component {16 NK linear terms (gconst at D), 16 NK
  // quadratic ones -- both in the frames' balanced units --, max |dgconst|, max |dlinear|_2, 0, 0}
  constexpr int TAB = 32 * NK + FB_FXW_ANCHORS * (32 * NK + 4);
  float *tab = lds + SLOTB4 * 4;
  constexpr int TABH = FB_FXW_ANCHORS * 16 * NK;  // + the anchors' f16 copies, two per float
  for (int i = tid; i < TAB + TABH; i += 256) tab[i] = g.anchor[i];
  __syncthreads();
  FXW_STAMP(1);
  float amax[2] = {1.0f, 1.0f};  // per frame: largest balanced |x|, x^2 (this lane's half of the dimensions)
  // one group of eight dimensions of the two frames at a time, straight into the f16 fragments.  down != 1: the range
  // shift's second pass (rare: reads the features again)
  auto load_frames = [&](const float (&down)[2], const bool first) {
#pragma unroll
    for (int c = 0; c < NK; ++c) {
      const int d0 = 16 * c + 8 * h;
      const float4 *sx = reinterpret_cast<const float4 *>(tab + d0), *sq = reinterpret_cast<const float4 *>(tab + 16 * NK + d0);
      const float4 x0 = sx[0], x1 = sx[1], y0 = sq[0], y1 = sq[1];
      const float xw[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w}, yw[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bool ok = rows[hf] < n_rows;
        float v[8], q[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d = d0 + 4 * u;
          float4 t = ft[c][hf][u];
          if (!first) t = *reinterpret_cast<const float4 *>(feats + (size_t)(ok ? rows[hf] : 0) * g.D + min(d, g.D - 4));
          // (a row past the end of the batch repeats row 0, its results are not stored; NK chunks: 16 (NK - 1) <= D)
          const bool in = c < NK - 1 || d < g.D;
          v[4 * u + 0] = in ? t.x : 0.0f; v[4 * u + 1] = in ? t.y : 0.0f;
          v[4 * u + 2] = in ? t.z : 0.0f; v[4 * u + 3] = in ? t.w : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = __fmul_rn(v[i], v[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (d0 + i >= g.D && d0 + i < g.D + 3) ? 1.0f : v[i];  // against the constants' three terms
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = __fmul_rn(v[i], xw[i]); q[i] = __fmul_rn(q[i], yw[i]); }  // exact
        if (first) {
#pragma unroll
          for (int i = 0; i < 8; ++i) amax[hf] = fmaxf(amax[hf], fmaxf(fabsf(v[i]), q[i]));
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) { v[i] = __fmul_rn(v[i], down[hf]); q[i] = __fmul_rn(q[i], down[hf]); }
        }
        fb_split2_frag(v, bx1[hf][c], bx2[hf][c]);
        fb_split2_frag(q, bq1[hf][c], bq2[hf][c]);
        // only the base items use these: parked in accumulation registers, read from there by their MFMAs
        asm volatile("" : "+a"(bx2[hf][c]), "+a"(bq1[hf][c]), "+a"(bq2[hf][c]));
      }
      __builtin_amdgcn_sched_barrier(0);  // one group at a time
    }
  };
  const float one2[2] = {1.0f, 1.0f};
  load_frames(one2, true);
  FXW_STAMP(2);
  // Range shift, per FRAME: a frame whose balanced values leave f16's range (|x| >= 181 spreads) is scaled down by a
  // power of two of its own, its accumulators hold t 2^-sh, and the whole wave takes the rescue path for every update
  // (`slow`), which multiplies each lane's values back -- by exactly 1 for the wave's other frames.
  float up[2] = {1.0f, 1.0f};
  bool any_shift = false;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) amax[hf] = fmaxf(amax[hf], __shfl_xor(amax[hf], 32, 64));
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(amax[0] >= 32768.0f || amax[1] >= 32768.0f) != 0ull, 0)) {
    float down[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int ex = (int)((__float_as_uint(amax[hf]) >> 23) & 0xffu) - 127;
      const int sh = amax[hf] >= 32768.0f ? min(ex - 14, 100) : 0;
      down[hf] = fb_pow2f(-sh);
      up[hf] = fb_pow2f(sh);
    }
    load_frames(down, false);
    any_shift = true;
  }
  // The anchors' log2-likelihoods of the two frames, from the LEADING f16 term of the balanced frame operand (a lower
  // bound with 64 to spare does not need more than its 11 bits; in the fragment pass the extra live values made hipcc
  // spill ~85 registers -- 23 us of scratch traffic per workgroup, measured).  Each lane sums its 8 of every 16
  // dimensions; |x|^2 in the same units for the Cauchy-Schwarz slack.
  float lbest[2] = {-3.0e38f, -3.0e38f};
  {
    // (packed f16: x'^2 by v_pk_mul_f16 -- below 32768 in a wave without range shift --, the sums by v_dot2_f32_f16 with
    //  f32 accumulation; the tables' f16 copies stand behind their f32 ones: tab + TAB, fb_load_gmm's layout in halves)
    const _Float16 *tabh = reinterpret_cast<const _Float16 *>(tab + TAB);
    f16x2 xs[2][NK][4];  // x'^2, zero past D
    float xx[2] = {0.0f, 0.0f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int c = 0; c < NK; ++c)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned xw_ = bx1[hf][c][u];  // (a copy first: __builtin_bit_cast of a vector ELEMENT reads element 0, hipcc 7.2)
          const f16x2 x = __builtin_bit_cast(f16x2, xw_);
          f16x2 q = x * x;
          const int d = 16 * c + 8 * h + 2 * u;
          if (c == NK - 1) { if (d >= g.D) q[0] = (_Float16)0.0f; if (d + 1 >= g.D) q[1] = (_Float16)0.0f; }
          xs[hf][c][u] = q;
          xx[hf] = __builtin_amdgcn_fdot2(x, x, xx[hf], false);
        }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {  // (the three constants' places hold 1.0: not part of |x|^2)
      const int lo = 16 * (NK - 1) + 8 * h;
      float ones = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ones += (lo + i >= g.D && lo + i < g.D + 3) ? 1.0f : 0.0f;
      xx[hf] -= ones;
    }
#pragma unroll
    for (int a = 0; a < FB_FXW_ANCHORS; ++a) {
      const float *at = tab + 32 * NK + a * (32 * NK + 4);
      const _Float16 *ah = tabh + a * (32 * NK);
      float lb[2] = {0.0f, 0.0f};
#pragma unroll
      for (int c = 0; c < NK; ++c) {
        const int d0 = 16 * c + 8 * h;
        const u32x4 lw = *reinterpret_cast<const u32x4 *>(ah + d0), qw = *reinterpret_cast<const u32x4 *>(ah + 16 * NK + d0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned lu = lw[u], qu = qw[u], xu = bx1[hf][c][u];
            lb[hf] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, lu), __builtin_bit_cast(f16x2, xu), lb[hf], false);
            lb[hf] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, qu), xs[hf][c][u], lb[hf], false);
          }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const float full = lb[hf] + __shfl_xor(lb[hf], 32, 64), x2 = fmaxf(xx[hf] + __shfl_xor(xx[hf], 32, 64), 0.0f);
        lbest[hf] = fmaxf(lbest[hf], full - at[32 * NK] - at[32 * NK + 1] * sqrtf(x2));
      }
    }
  }
  // the frames' reference R (see above fb_fxw_step): both lanes of a frame add the same two numbers.  A wave with the
  // range shift keeps R = 0 and takes the rescue path for every update (`slow`, uniform over the wave)
  const bool slow = any_shift;  // (the same in every lane: a scalar for the branches below)
  float rc[2], rp[2], rn[2];  // reference of the current tile's accumulators, of the previous tile's, proposed for the next
  // -R enters the quadratic item's accumulation through the K padding: the x^2 operand of frame j carries -(R mod 2048)
  // at K = D + 3 and -(R div 2048) at K = D + 4 (integers below 2049: exact in f16), the quadratic item's image 1 and
  // 2048 there (fb_load_gmm) -- no registers, no instructions.  The lanes that hold those K places patch their fragment.
  auto set_ref = [&](const int hf, const float R) {
    const float hi = truncf(__fmul_rn(R, 1.0f / 2048.0f)), lo = __fmaf_rn(hi, -2048.0f, R);
    u32x4 f = bq1[hf][NK - 1];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int k = g.D + 3 + sidx - 16 * (NK - 1), el = k & 7;
      const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(sidx == 0 ? -lo : -hi));
      if ((k >> 3) == h) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u == (el >> 1)) f[u] = (el & 1) ? ((f[u] & 0x0000ffffu) | (bits << 16)) : ((f[u] & 0xffff0000u) | bits);
      }
    }
    bq1[hf][NK - 1] = f;
    asm volatile("" : "+a"(bq1[hf][NK - 1]));
  };
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const float r0 = floorf(lbest[hf]) + FB_FXW_ROFF;
    rc[hf] = slow ? 0.0f : fminf(fmaxf(r0, -4.0e6f), 4.0e6f);  // (the two K places hold |R| < 2^22)
    rp[hf] = rc[hf];
    rn[hf] = rc[hf];
    set_ref(hf, rc[hf]);
  }
  const float r_init[2] = {rc[0], rc[1]};   // (what every component chunk of this workgroup starts from)
#pragma unroll
  for (int m = 0; m < 2 * M; ++m) { st_m[m * 256 + tid] = rc[m & 1]; st_s[m * 256 + tid] = 0.0f; }

  // ---- the frames' fp6 operands of the F6 tiles (fb_fxw_step6; block layout: fb_load_gmm), from the f16 fragments:
  //      block 0 = x_1 of the chunks 0 .. 3, block 1 = x_2 of the same places times 2^12 (by the scale alone), block 2 =
  //      {x_1 of chunk 4, x_2 of chunk 4 times 2^12 (exact: |x_2| <= 8 below the range shift), x_1 of chunk 4 again (for
  //      delta_2's second term), 8 zeros} in the lanes h = 0, zero in the others.  One scale per
  //      lane and block: the smallest power of two that brings the block's largest magnitude to <= 7.5.
  i32x8 fq[2][3];
  int fsc[2][3];
  const bool has_f6 = g.delta_t6 > g.delta_t3;
  if (has_f6) {
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto convert = [&](const u32x16 &wv, const int extra, i32x8 &out, int &sc) {
      u16x2 mx = {0, 0};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned av = wv[i] & 0x7fff7fffu;  // |f16| patterns order like unsigned integers
        mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, av));
      }
      const unsigned short top = mx[0] > mx[1] ? mx[0] : mx[1];
      const float mf = (float)__builtin_bit_cast(_Float16, top);
      // smallest e with mf 2^-e <= 7.5 (one more where mf / 7.5 is an exact power of two): exponent(mf / 7.5) + 1
      int byte = (int)((__float_as_uint(__fmul_rn(mf, 0.13333334f)) >> 23) & 0xffu) + 1;
      byte = byte > 240 ? 240 : byte;  // (inf / NaN frames: their scores are NaN anyway)
      const auto q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, wv), __uint_as_float((unsigned)byte << 23));
      out = i32x8{(int)q[0], (int)q[1], (int)q[2], (int)q[3], (int)q[4], (int)q[5], 0, 0};
      sc = byte + extra;
    };
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      u32x16 w0, w1, w2;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 4; ++u) { w0[4 * c + u] = bx1[hf][c][u]; w1[4 * c + u] = bx2[hf][c][u]; }
      const f16x2 k4096 = {(_Float16)4096.0f, (_Float16)4096.0f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // (lanes h = 1 hold the constants' places there: no dimensions, zeros)
        const unsigned xu = bx2[hf][NK - 1][u];
        const f16x2 up2 = __builtin_bit_cast(f16x2, xu) * k4096;
        w2[u] = h ? 0u : bx1[hf][NK - 1][u];
        w2[4 + u] = h ? 0u : __builtin_bit_cast(unsigned, up2);
        w2[8 + u] = w2[u];
        w2[12 + u] = 0u;
      }
      convert(w0, 0, fq[hf][0], fsc[hf][0]);
      convert(w1, 12, fq[hf][1], fsc[hf][1]);
      convert(w2, 0, fq[hf][2], fsc[hf][2]);
    }
  } else {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int b = 0; b < 3; ++b) { fq[hf][b] = i32x8{0, 0, 0, 0, 0, 0, 0, 0}; fsc[hf][b] = 127; }
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int b = 0; b < 3; ++b) asm volatile("" : "+a"(fq[hf][b]));  // parked beside the other frame operands

  for (int sub = 0; sub < n_sub; ++sub) {
  const int chunk_i = gchunk + sub * grid_chunks;
  if (sub > 0) {   // the state of a workgroup that has not seen a tile yet
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      rc[hf] = r_init[hf]; rp[hf] = r_init[hf]; rn[hf] = r_init[hf];
      set_ref(hf, rc[hf]);
    }
#pragma unroll
    for (int m = 0; m < 2 * M; ++m) { st_m[m * 256 + tid] = rc[m & 1]; st_s[m * 256 + tid] = 0.0f; }
  }
  // Component tiles of this chunk: chunk_i, chunk_i + n_chunks, ... (strided, so that every chunk gets the same mix of
  // the tile classes below).  fb_load_gmm stores the components SORTED by how far the other models moved them from
  // the base model (the order is free under logsumexp), so the products per K chunk a delta item needs (P, above
  // fb_fxw_step) fall from tile to tile: tiles [0, delta_t3) run P = 3, [delta_t3, delta_t2) P = 2, the rest P = 1.
  const int n_t = chunk_i < g.n_tiles ? (g.n_tiles - chunk_i + n_chunks - 1) / n_chunks : 0;
  auto tiles_below = [&](int bound) {  // how many of this chunk's tiles have an index < bound
    const int c = bound > chunk_i ? (bound - chunk_i + n_chunks - 1) / n_chunks : 0;
    return c < n_t ? c : n_t;
  };
  const int n_p3 = tiles_below(g.delta_t3), n_p36 = tiles_below(g.delta_t6), n_p32 = tiles_below(g.delta_t2);
  const u32x4 *gimg = g.images_fd;
  auto tile_of = [&](int t) { return chunk_i + (t < n_t - 1 ? t : n_t - 1) * n_chunks; };  // (clamped: see srcA below)

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
  // after a step that carried a fast update: the guard -- the new sums' bit patterns, as unsigned integers, against
  // 2^100's (+ 1: inf, NaN and anything negative lie above it; 0 for a wave on the rescue path: always) -- and the
  // rescue of the lanes that fail it (cold)
  const unsigned guard = slow ? 0u : __float_as_uint(FB_FXW_SMAX) + 1u;
  auto settle = [&](const f32x16 &p0, const f32x16 &p1, float *pm, float *ps, const float (&rt)[2], const FbFxwUpd &u) {
    const unsigned worst = max(__float_as_uint(u.sn0), __float_as_uint(u.sn1));
    FXW_COUNT(0);
#ifdef FB_FXW_COUNT
    {  // how many of the update's 16 slices (a register pair = two components x the wave's 64 frames) could have been
       // skipped wave-uniformly: every value more than 25 log2 units below the frame's running sum -- Kaldi's LogSumExp
       // drops what lies log2(1 / FLT_EPSILON) = 23 below the maximum (task 3b: measured)
      const float th0 = __builtin_amdgcn_logf(u.so0) - 25.0f, th1 = __builtin_amdgcn_logf(u.so1) - 25.0f;  // v_log_f32 = log2
      int n = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (__builtin_amdgcn_ballot_w64(!(p0[r] < th0) || !(p1[r] < th1)) == 0ull) ++n;
      if ((threadIdx.x & 63) == 0) atomicAdd(&g_fxw_counts[3], (unsigned long long)n);
    }
#endif
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(worst >= guard) != 0ull, 0)) {
      FXW_COUNT(1);
      fb_fxw_slow_update(p0, up[0], rt[0], pm[0], u.so0, pm, ps, rn[0], __float_as_uint(u.sn0) >= guard);
      fb_fxw_slow_update(p1, up[1], rt[1], pm[256], u.so1, pm + 256, ps + 256, rn[1], __float_as_uint(u.sn1) >= guard);
    }
  };
  auto update = [&](const f32x16 &v0, const f32x16 &v1, int model) {  // the kernel's tail: every lane the classical way
    float *pm = st_m + (2 * model) * 256 + tid, *ps = st_s + (2 * model) * 256 + tid;
    fb_fxw_slow_update(v0, up[0], rc[0], pm[0], ps[0], pm, ps, rn[0], true);
    fb_fxw_slow_update(v1, up[1], rc[1], pm[256], ps[256], pm + 256, ps + 256, rn[1], true);
  };
  auto publish = [&]() {  // everything this wave asked for has landed; the barrier publishes all four waves' pieces
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // The base steps of the FIRST tile have no finished models to carry, and a branch for them costs more than a bogus
  // update (nothing hides an instruction fetch at one wave per SIMD): they "update" the last two models with 16 values
  // -1e30, whose exponentials are zero.
  f32x16 hq[2], acc[2][2], zero;  // acc[set][half]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    hq[0][r] = 0.f; hq[1][r] = 0.f; zero[r] = 0.f;
    acc[0][0][r] = -1.0e30f; acc[0][1][r] = -1.0e30f; acc[1][0][r] = -1.0e30f; acc[1][1][r] = -1.0e30f;
  }
  FbFxwUpd uu;
  float wd[2] = {1.0f, 1.0f};  // 2^(rp - rc): the factor of the two deferred updates behind a moved reference
  FXW_STAMP(3);
  fb_fxw_fetch<GA, NPIECE>(gimg + (size_t)tile_of(0) * NI * IMG4 + lane, ring_lds, wv);  // group A of the first tile
  publish();
  FXW_STAMP(4);
  u32x4 z1, z2;  // chunk 0 of the item in front (fb_fxw_step)
  constexpr int PW_A = (GA * NPIECE + 3) / 4, PW_B = (GB * NPIECE + 3) / 4;  // LDS-DMA pieces per wave for a group
  // one tile with P products per K chunk in its delta items (a compile-time constant: the tile is straight-line code)
  auto tile = [&](auto pc, const int t) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    // a rescue of the last tile proposed a new reference for some frame: from this tile on (the two deferred updates
    // below still belong to the old one, rp)
    rp[0] = rc[0]; rp[1] = rc[1];
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(rn[0] > rc[0] || rn[1] > rc[1] || wd[0] != 1.0f || wd[1] != 1.0f) != 0ull, 0)) {
      FXW_COUNT(2);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {  // the two lanes of a frame (its rows 0 .. 15 / 16 .. 31 of the tile) agree on the larger proposal
        rn[hf] = fminf(fmaxf(rn[hf], __shfl_xor(rn[hf], 32, 64)), 4.0e6f);
        rc[hf] = slow ? 0.0f : rn[hf];
        set_ref(hf, rc[hf]);
        // the fast updates add to sums that are relative to rc: every model's state moves to the new reference (a model
        // the rescue re-referenced on its own comes back too), and the two deferred updates below, whose values are
        // relative to rp, take the factor 2^(rp - rc) -- 1 again from the next boundary on
        wd[hf] = __builtin_amdgcn_exp2f(rp[hf] - rc[hf]);
        if (!slow) {
#pragma unroll
          for (int m = 0; m < M; ++m) {
            float *pm = st_m + (2 * m + hf) * 256 + tid, *ps = st_s + (2 * m + hf) * 256 + tid;
            *ps = *ps * __builtin_amdgcn_exp2f(*pm - rc[hf]);
            *pm = rc[hf];
          }
        }
      }
    }
    // this wave's share of the groups requested during this tile: group B of this tile (while A runs), group A of the
    // chunk's next one (while B runs); past the chunk's end the last tile's group A is read again and never used
    const u32x4 *srcB = gimg + ((size_t)tile_of(t) * NI + GA) * IMG4 + (size_t)wv * (PW_B * 64) + lane;
    const u32x4 *srcA = gimg + (size_t)tile_of(t + 1) * NI * IMG4 + (size_t)wv * (PW_A * 64) + lane;
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
          fb_fxw_step<NK, 3, true, true, true>(cur4, nxt4, pf0, lane, z1, z2, bq1, bq2, zero, zero, hq[0], hq[1], acc[(M - 2) & 1][0], acc[(M - 2) & 1][1], ps, wd[0], wd[1], uu, dsrc, ddst, dq0, dn);
          settle(acc[(M - 2) & 1][0], acc[(M - 2) & 1][1], pm, ps, rp, uu);
        } else {
          fb_fxw_step<NK, 3, false, true>(cur4, nxt4, pf0, lane, z1, z2, bq1, bq2, zero, zero, hq[0], hq[1], zero, zero, st_s, 1.f, 1.f, uu, dsrc, ddst, dq0, dn);
        }
      } else if (jj == 1) {
        float *pm = st_m + (2 * (M - 1)) * 256 + tid, *ps = st_s + (2 * (M - 1)) * 256 + tid;
        fb_fxw_step<NK, 3, true, false, true>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], hq[0], hq[1], acc[(M - 1) & 1][0], acc[(M - 1) & 1][1], ps, wd[0], wd[1], uu, dsrc, ddst, dq0, dn);
        settle(acc[(M - 1) & 1][0], acc[(M - 1) & 1][1], pm, ps, rp, uu);
      } else if (jj == 2) {
        float *pm = st_m + tid, *ps = st_s + tid;
        if constexpr (P == 6)
          fb_fxw_step6<NK, true>(cur4, nxt4, pf0, lane, z1, bx1, fq, fsc, hq[0], hq[1], acc[1][0], acc[1][1], hq[0], hq[1], ps, uu, dsrc, ddst, dq0, dn);
        else
          fb_fxw_step<NK, P, true>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[1][0], acc[1][1], hq[0], hq[1], ps, 1.f, 1.f, uu,
                                   dsrc, ddst, dq0, dn);
        settle(hq[0], hq[1], pm, ps, rc, uu);
      } else if (DEFER && jj == NI - 1) {
        if constexpr (P == 6)
          fb_fxw_step6<NK, false>(cur4, nxt4, pf0, lane, z1, bx1, fq, fsc, hq[0], hq[1], acc[(jj - 1) & 1][0], acc[(jj - 1) & 1][1],
                                  zero, zero, st_s, uu, dsrc, ddst, dq0, dn);
        else
          fb_fxw_step<NK, P, false>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[(jj - 1) & 1][0], acc[(jj - 1) & 1][1],
                                    zero, zero, st_s, 1.f, 1.f, uu, dsrc, ddst, dq0, dn);
      } else {
        float *pm = st_m + (2 * (jj - 2)) * 256 + tid, *ps = st_s + (2 * (jj - 2)) * 256 + tid;
        if constexpr (P == 6)
          fb_fxw_step6<NK, true>(cur4, nxt4, pf0, lane, z1, bx1, fq, fsc, hq[0], hq[1], acc[(jj - 1) & 1][0], acc[(jj - 1) & 1][1],
                                 acc[(jj - 2) & 1][0], acc[(jj - 2) & 1][1], ps, uu, dsrc, ddst, dq0, dn);
        else
          fb_fxw_step<NK, P, true>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[0], hq[1], acc[(jj - 1) & 1][0], acc[(jj - 1) & 1][1],
                                   acc[(jj - 2) & 1][0], acc[(jj - 2) & 1][1], ps, 1.f, 1.f, uu, dsrc, ddst, dq0, dn);
        settle(acc[(jj - 2) & 1][0], acc[(jj - 2) & 1][1], pm, ps, rc, uu);
      }
      if (jj == GA - 1 || jj == NI - 1) publish();
    }
  };
  {
    int t = 0;
    for (; t < n_p3; ++t) tile(std::integral_constant<int, 3>{}, t);
    for (; t < n_p36; ++t) tile(std::integral_constant<int, 6>{}, t);
    for (; t < n_p32; ++t) tile(std::integral_constant<int, 2>{}, t);
    for (; t < n_t; ++t) tile(std::integral_constant<int, 1>{}, t);
  }
  FXW_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the updates the last tile deferred: the fast form (one exponential and one addition per value, the guard, the cold
  // rescue) without MFMAs to thread it through -- the classical form here (a maximum, a re-reference, fma + exp + add
  // per value) was 2.3 us of every workgroup's 58
  auto tail_update = [&](const f32x16 &v0, const f32x16 &v1, int model) {
    if (slow) { update(v0, v1, model); return; }
    float *pm = st_m + (2 * model) * 256 + tid, *ps = st_s + (2 * model) * 256 + tid;
    FbFxwUpd u;
    u.so0 = ps[0]; u.so1 = ps[256];
    float e0 = __builtin_amdgcn_exp2f(v0[0]), d0 = __builtin_amdgcn_exp2f(v0[1]);
    float e1 = __builtin_amdgcn_exp2f(v1[0]), d1 = __builtin_amdgcn_exp2f(v1[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) {
      e0 = __fadd_rn(e0, __builtin_amdgcn_exp2f(v0[r])); d0 = __fadd_rn(d0, __builtin_amdgcn_exp2f(v0[r + 1]));
      e1 = __fadd_rn(e1, __builtin_amdgcn_exp2f(v1[r])); d1 = __fadd_rn(d1, __builtin_amdgcn_exp2f(v1[r + 1]));
    }
    u.sn0 = __fadd_rn(__fadd_rn(e0, d0), u.so0);
    u.sn1 = __fadd_rn(__fadd_rn(e1, d1), u.so1);
    ps[0] = u.sn0; ps[256] = u.sn1;
    settle(v0, v1, pm, ps, rc, u);
  };
  if constexpr (M >= 3) tail_update(acc[(M - 2) & 1][0], acc[(M - 2) & 1][1], M - 2);
  tail_update(acc[(M - 1) & 1][0], acc[(M - 1) & 1][1], M - 1);
  FXW_STAMP(6);

#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      // (mref, s) in log2 units -> the (m, sum exp(ll - m)) convention of the chunk merge / k_gmm_finalize:
      // m = fl(mref ln 2); what the rounding leaves, d = mref ln 2 - m (|d| < 1e-4: the product's exact residual from
      // an fma + mref times the constant's own error), is folded into s: s e^d = s (1 + d + d^2 / 2)
      const float ms = st_m[(2 * m + hf) * 256 + tid], s0 = st_s[(2 * m + hf) * 256 + tid];
      const float ln2f = 0.693147182464599609375f;                 // fl(ln 2); ln 2 - fl(ln 2) = -1.9046543e-9
      const float mm = __fmul_rn(ms, ln2f);
      const float dd = __fmaf_rn(ms, -1.9046542999e-9f, __fmaf_rn(ms, ln2f, -mm));
      const float ss = __fmaf_rn(s0, __fmaf_rn(__fmul_rn(0.5f, dd), dd, dd), s0);
      const float m2 = __shfl_xor(mm, 32, 64), s2 = __shfl_xor(ss, 32, 64);
      const float mx = fmaxf(mm, m2);
      const float sx = ss * __expf(mm - mx) + s2 * __expf(m2 - mx);
      if (h == 0 && rows[hf] < n_rows) {
        const int mg = m == 0 ? 0 : g.pass_first + m - 1;  // this pass's model m among the g.M the partials are laid out for
        const size_t o = ((size_t)chunk_i * g.M + mg) * rows_cap + rows[hf];
        part_m[o] = mx;
        part_s[o] = sx;
      }
    }
  }
  FXW_STAMP(7);
}


// ---------------------------------------------------------------------------------------------
// k_gsel_w: the matrix-core passes of the threshold gselect (gmm_kernels.hip, above fb_launch_gsel; round 6) in this file's
// form -- one wave per SIMD, 64 frames per wave as two independent 32-frame chains that share every parameter fragment,
// the parameter items by LDS-DMA into two slots of TS = 2 tiles each (one workgroup barrier per two tiles), the frame
// operands' second terms parked in accumulation registers.  k_gmm_fx2_sel (four waves of 32 frames, a barrier and a pass
// through staging registers per item) took 39 + 43 us for the two passes at configs[2] size against ~16 us of MFMA time
// each.  The MFMAs of a half run on the same operands in the same order as fb_fx_step's (a2 b1, a1 b2, a1 b1 per K chunk, the
// base item continuing the quadratic one's accumulator), so a value has the bits the dump would have stored -- in the
// accumulators' own scale: a frame's values all carry the same power of two 2^(kacc - sh), which no maximum, comparison or
// ranking cares about, so nothing is un-scaled anywhere.
//   PICK = false  pass A: a lane's maximum of its 16 values -> gmax[row][2 n_tiles] (through LDS, whole 128-byte runs)
//   PICK = true   pass B: a lane whose maximum reaches tau(row) stores its 16 values as one 64-byte record
//                 gval[((row n_chunks + chunk) capc + k) 16 ..], k from the row's counter in LDS, and the group's id
//                 (2 tile_in_chunk + h) beside it; capc = 2 tiles_per_chunk records per (row, chunk): every group fits, there
//                 is no overflow and no rescue.  k_gsel_final_w ranks the values >= tau of those records.
// Shapes: NK = 5 (D = 72), C a multiple of 32, tiles per chunk a multiple of 4 and at most 128; everything else stays on
// k_gmm_fx2_sel.
template <int NK, bool PICK>
__global__ __launch_bounds__(256, 1) void k_gsel_w(FbGmmDev g, const float *__restrict__ feats, const int *__restrict__ n_rows_ptr,
                                                   int tiles_per_chunk, int n_chunks, float *__restrict__ gmax,
                                                   const float *__restrict__ tau, float *__restrict__ gval,
                                                   unsigned char *__restrict__ gid, int *__restrict__ gcnt, int xcd_map,
                                                   int tiles_a) {
  if (g.stop && *g.stop) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IMG4 = 2 * NK * 64;      // 16-byte units per item
  constexpr int TS = 2, NIG = 2 * TS;    // tiles / items per slot
  constexpr int SLOT4 = NIG * IMG4;      // 16-byte units per slot
  constexpr int NPIECE = IMG4 / 64;      // 1 KB pieces per item
  constexpr int PW = NIG * NPIECE / 4;   // pieces per wave and slot
  static_assert((NIG * NPIECE) % 4 == 0, "a slot is dealt over four waves");
  FXW_STAMP(8);
  const int n_rows = *n_rows_ptr;
  int strip_i, chunk_i;
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
  const u32x4 *slot0 = reinterpret_cast<const u32x4 *>(lds);
  float *s_mx = lds + 2 * SLOT4 * 4;                                  // pass A: [4 waves][16 tiles][2 halves][64 lanes]
  int *s_cnt = reinterpret_cast<int *>(lds + 2 * SLOT4 * 4);          // pass B: [256 rows]
  unsigned char *s_gid = reinterpret_cast<unsigned char *>(s_cnt + 256);  //   [256 rows][capc]
  const int capc = 2 * tiles_per_chunk;

  u32x4 bx1[2][NK], bx2[2][NK], bq1[2][NK], bq2[2][NK];
  int rows[2];
  float4 ft[2][NK][2];   // both frames' loads go out before the first is used
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    rows[hf] = strip0 + w * 64 + hf * 32 + j;
    fb_fx_frame_load<NK>(g, feats, rows[hf], n_rows, h, ft[hf]);
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    (void)fb_fx_frame_make<NK, true>(g, feats, ft[hf], rows[hf], n_rows, h, bx1[hf], bx2[hf], bq1[hf], bq2[hf]);
#pragma unroll
    for (int c = 0; c < NK; ++c) asm volatile("" : "+a"(bx2[hf][c]), "+a"(bq1[hf][c]), "+a"(bq2[hf][c]));
    __builtin_amdgcn_sched_barrier(0);  // one half at a time
  }
  FXW_STAMP(9);
  float tauv[2] = {FLT_MAX, FLT_MAX};
  float4 *gb[2] = {nullptr, nullptr};
  if constexpr (PICK) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (rows[hf] < n_rows) tauv[hf] = tau[rows[hf]];
      gb[hf] = reinterpret_cast<float4 *>(gval) + ((size_t)(rows[hf] < n_rows ? rows[hf] : 0) * n_chunks + chunk_i) * capc * 4;
    }
    s_cnt[tid] = 0;
  }
  const int tile0 = chunk_i * tiles_per_chunk;
  // pass A may look at the first tiles_a tiles of the chunk only (a multiple of 4, like tiles_per_chunk): the nsel-th largest
  // group maximum of ANY subset of the components is a lower bound of the row's nsel-th largest value -- half the components
  // give a tau near the 2 nsel-th largest value, twice the records in pass B and half of pass A's matrix work
  const int n_tiles_run = PICK ? tiles_per_chunk : tiles_a;
  const int n_grp = n_tiles_run / TS;                                     // slot contents of this pass (even: the launcher)
  const u32x4 *gimg = g.images_fx + (size_t)tile0 * 2 * IMG4;
  const int wv = __builtin_amdgcn_readfirstlane(w);
  const unsigned ring_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)lds;
  auto publish = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // two accumulator sets, tiles alternate: what a tile leaves is looked at (epilogue) behind the NEXT tile's quadratic step --
  // its values are long complete there, and the vector work runs while the matrix pipe finishes that step's last products
  f32x16 hq[2][2], zero;   // [tile parity][frame half]
#pragma unroll
  for (int r = 0; r < 16; ++r) { hq[0][0][r] = 0.f; hq[0][1][r] = 0.f; hq[1][0][r] = 0.f; hq[1][1][r] = 0.f; zero[r] = 0.f; }
  FbFxwUpd uu;
  FXW_STAMP(10);
  fb_fxw_fetch<NIG, NPIECE>(gimg + lane, ring_lds, wv);
  publish();
  FXW_STAMP(11);
  u32x4 z1, z2;
  // the tile's epilogue: what leaves the accumulators (tl = tile within the chunk)
  auto epilogue = [&](const int tl, const int par) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const f32x16 &v = hq[par][hf];
      float m = fmaxf(fmaxf(v[0], v[1]), v[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, v[r]), v[r + 1]);
      m = fmaxf(m, v[15]);
      if constexpr (!PICK) {
        s_mx[((w * 16 + (tl & 15)) * 2 + hf) * 64 + lane] = m;
      } else {
        if (m >= tauv[hf]) {
          const int rl = w * 64 + hf * 32 + j;
          const int k = atomicAdd(&s_cnt[rl], 1);   // < capc: a (tile, h) group arrives once
          float4 *dst = gb[hf] + (size_t)k * 4;
          dst[0] = make_float4(v[0], v[1], v[2], v[3]);
          dst[1] = make_float4(v[4], v[5], v[6], v[7]);
          dst[2] = make_float4(v[8], v[9], v[10], v[11]);
          dst[3] = make_float4(v[12], v[13], v[14], v[15]);
          s_gid[rl * capc + k] = (unsigned char)(2 * tl + h);
        }
      }
    }
    if constexpr (!PICK) {
      if ((tl & 15) == 15 || tl == tiles_a - 1) {
        // this wave's 64 rows x (up to) 16 tiles x 2 groups, written as whole runs of the rows of gmax
        const int nt = (tl & 15) + 1, per_row = 2 * nt, NGr = 2 * n_chunks * tiles_a, pos0 = 2 * (chunk_i * tiles_a + (tl & ~15));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < 64 * per_row; i += 64) {
          const int r2 = i / per_row, pos = i - r2 * per_row;
          const int rg = strip0 + w * 64 + r2;
          if (rg < n_rows) gmax[(size_t)rg * NGr + pos0 + pos] = s_mx[((w * 16 + (pos >> 1)) * 2 + (r2 >> 5)) * 64 + (pos & 1) * 32 + (r2 & 31)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  };
  for (int gp = 0; gp < n_grp; gp += 2) {   // slot A = group gp, slot B = group gp + 1
    const u32x4 *srcB = gimg + (size_t)min(gp + 1, n_grp - 1) * SLOT4 + (size_t)wv * (PW * 64) + lane;
    const u32x4 *srcA = gimg + (size_t)min(gp + 2, n_grp - 1) * SLOT4 + (size_t)wv * (PW * 64) + lane;
    const unsigned dstB = ring_lds + SLOT4 * 16 + (unsigned)wv * (PW * 1024), dstA = ring_lds + (unsigned)wv * (PW * 1024);
#pragma clang loop unroll(full)
    for (int jj = 0; jj < 2 * NIG; ++jj) {
      const bool in_b = jj >= NIG;
      const int gs = jj % NIG;   // step within the slot
      const u32x4 *cur4 = slot0 + (in_b ? SLOT4 : 0) + gs * IMG4;
      const u32x4 *nxt4 = slot0 + (in_b ? SLOT4 : 0) + (gs + 1 < NIG ? gs + 1 : 0) * IMG4;
      const bool pf0 = gs + 1 < NIG;
      if (gs == 0) {
        z1 = cur4[(0 * NK + 0) * 64 + lane];
        z2 = cur4[(1 * NK + 0) * 64 + lane];
      }
      const u32x4 *dsrc = in_b ? srcA : srcB;
      const unsigned ddst = in_b ? dstA : dstB;
      const int dq0 = 5 * gs;
      int dn = dq0 < PW ? (PW - dq0 < 5 ? PW - dq0 : 5) : 0;
      if (gs == 0) {
#pragma unroll
        for (int q = 0; q < dn; ++q) fb_glds16(dsrc + q * 64, ddst + (unsigned)q * 1024u);
        dn = 0;
      }
      constexpr int dummy = 0; (void)dummy;
      const int par = (jj >> 1) & 1;                       // this tile's accumulator set (TS = 2: tiles per loop pass = 4)
      const int tl = gp * TS + (jj >> 1);                  // tile within the chunk
      if (!(jj & 1)) {
        fb_fxw_step<NK, 3, false, true>(cur4, nxt4, pf0, lane, z1, z2, bq1, bq2, zero, zero, hq[par][0], hq[par][1], zero, zero, lds, 1.f, 1.f,
                                        uu, dsrc, ddst, dq0, dn);
        if (tl > 0) epilogue(tl - 1, par ^ 1);             // the previous tile (its base step is one step back)
      } else {
        fb_fxw_step<NK, 3, false, false>(cur4, nxt4, pf0, lane, z1, z2, bx1, bx2, hq[par][0], hq[par][1], hq[par][0], hq[par][1], zero, zero, lds,
                                         1.f, 1.f, uu, dsrc, ddst, dq0, dn);
      }
      if (gs == NIG - 1) publish();
    }
  }
  epilogue(n_tiles_run - 1, 1);   // the last tile (a multiple of 4 tiles: odd parity)
  FXW_STAMP(12);
  if constexpr (PICK) {
    __syncthreads();   // every wave's last append
    for (int i = tid; i < 256 * (capc / 4); i += 256) {   // the group ids, 4 at a time (capc is a multiple of 8)
      const int r2 = i / (capc / 4), q = i - r2 * (capc / 4);
      const int rg = strip0 + r2;
      if (rg < n_rows)
        reinterpret_cast<unsigned *>(gid + ((size_t)rg * n_chunks + chunk_i) * capc)[q] = reinterpret_cast<const unsigned *>(s_gid + r2 * capc)[q];
    }
    if (strip0 + tid < n_rows) gcnt[(size_t)(strip0 + tid) * n_chunks + chunk_i] = s_cnt[tid];
  }
  FXW_STAMP(13);
}

bool fb_gsel_w_applies(const FbGmmDev &g, int n_chunks) {
  const bool off = getenv("FB_GSEL_NARROW") != nullptr;   // A/B and tests: k_gmm_fx2_sel (read per batch)
  if (off || g.NKF != 5 || (g.C & 31) != 0 || (g.D & 3) != 0 || g.n_tiles % n_chunks != 0) return false;
  const int tpc = g.n_tiles / n_chunks;
  return (tpc & 3) == 0 && tpc <= 128;
}
// one pass of the wide threshold gselect (pick = 0: group maxima, 1: the records of the groups that reach tau)
// tiles per chunk pass A looks at (fb_launch_gsel_w with pick = 0 writes gmax rows of 2 n_chunks fb_gsel_w_tiles_a() floats):
// all of them; FB_GSEL_A_HALF=1: half of them when that is still a multiple of 4 and leaves at least 3 nsel groups -- measured at
// configs[2] size: pass A 34.2 -> 22.0 us, k_gsel_tau 7.8 -> 6.1, but 42 records per row instead of 20: pass B 32.8 -> 36.5,
// k_gsel_final_w 10.3 -> 20.6 (it reads the records back): 85.2 against 85.1 us -- not the default
int fb_gsel_w_tiles_a(const FbGmmDev &g, int n_chunks, int nsel) {
  const int tpc = g.n_tiles / n_chunks;
  const bool full = getenv("FB_GSEL_A_HALF") == nullptr;
  if (!full && (tpc & 7) == 0 && n_chunks * tpc >= 3 * nsel) return tpc / 2;   // (2 groups per tile: n_chunks tpc groups)
  return tpc;
}
void fb_launch_gsel_w(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap, int n_chunks, int pick,
                      float *gmax, const float *tau, float *gval, unsigned char *gid, int *gcnt, int tiles_a) {
  constexpr int NK = 5;
  const int strips = (rows_cap + 255) / 256, tpc = g.n_tiles / n_chunks;
  dim3 grid((unsigned)strips, (unsigned)n_chunks);
  int xcd_map = 0;
  static const bool no_xcd_map = getenv("FB_GMM_NO_XCD_MAP") != nullptr;
  if ((n_chunks == 1 || n_chunks == 2 || n_chunks == 4 || n_chunks == 8) && !no_xcd_map) {
    const int per = 8 / n_chunks;
    grid = dim3((unsigned)(8 * ((strips + per - 1) / per)), 1);
    xcd_map = n_chunks;
  }
  const size_t slots = (size_t)2 * 4 * (2 * NK * 64) * 16;
  const size_t ldsb = slots + (pick ? sizeof(int) * 256 + (size_t)256 * 2 * tpc : sizeof(float) * 4 * 16 * 2 * 64);
  static std::atomic<unsigned long long> optin{0};
  unsigned long long bit = 0;
  if (ldsb > 64 * 1024 && fb_device_needs_optin(optin, &bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gsel_w<NK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_gsel_w<NK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess)
      optin.fetch_or(bit, std::memory_order_release);
  }
  if (pick)
    hipLaunchKernelGGL((k_gsel_w<NK, true>), grid, dim3(256), ldsb, s, g, feats, n_rows_ptr, tpc, n_chunks, gmax, tau, gval, gid, gcnt, xcd_map, tiles_a);
  else
    hipLaunchKernelGGL((k_gsel_w<NK, false>), grid, dim3(256), ldsb, s, g, feats, n_rows_ptr, tpc, n_chunks, gmax, tau, gval, gid, gcnt, xcd_map, tiles_a);
}

template <int NK, int M>
static void launch_gmm_fxw_t(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr,
                             int rows_cap, int n_chunks, float *part_m, float *part_s) {
  const int strips = (rows_cap + 255) / 256;
  // (component chunks a workgroup scores one after the other: FbGmmDev::fxw_sub when it divides the chunk count)
  FbGmmDev gl = g;
  gl.fxw_sub = (g.fxw_sub > 1 && n_chunks % g.fxw_sub == 0) ? g.fxw_sub : 1;
  const int gchunks = n_chunks / gl.fxw_sub;
  dim3 grid((unsigned)strips, (unsigned)gchunks);
  int xcd_map = 0;
  static const bool no_xcd_map = getenv("FB_GMM_NO_XCD_MAP") != nullptr;
  if ((gchunks == 1 || gchunks == 2 || gchunks == 4 || gchunks == 8) && !no_xcd_map) {
    const int per = 8 / gchunks;
    grid = dim3((unsigned)(8 * ((strips + per - 1) / per)), 1);
    xcd_map = gchunks;
  }
  const size_t ldsb = ((size_t)(M + 1) * 2 * NK * 64 + 2 * 192) * 16 + (size_t)2 * M * 512 * sizeof(float);  // one tile (two padded slots) + the state
  static std::atomic<unsigned long long> optin{0};
  unsigned long long bit = 0;
  if (ldsb > 64 * 1024 && fb_device_needs_optin(optin, &bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gmm_fx2w<NK, M>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) == hipSuccess)
      optin.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_gmm_fx2w<NK, M>), grid, dim3(256), ldsb, s, gl, feats, n_rows_ptr, n_chunks, rows_cap, part_m, part_s,
                     xcd_map);
}
// k_gmm_fx2w is instantiated for the shapes the reference's systems have with the recipe's 72-dimensional features
// (NKF = 5): one variance group, every component tile full, 2 <= M <= FB_FXW_MAX_M models per launch (SV: UBM + 1; OSI: UBM +
// up to 9 speakers; CSI: up to 10 speakers; larger sites in up to FB_FXW_MAX_PASS passes).  Everything else runs on k_gmm_fx2.
bool fb_gmm_use_wide(const FbGmmDev &g) {
  const bool off = getenv("FB_GMM_NARROW") != nullptr;  // read per call: the tests switch it inside one process
  return g.mode == FB_GMM_MODE_FX2 && !off && g.NKF == 5 && (g.D & 3) == 0 && g.n_items == g.M + 1 && (g.C & 31) == 0 && g.M >= 2 &&
         g.n_pass >= 1 && g.item_model_host_q_first && g.delta_p >= 1 && g.images_fd != nullptr && g.anchor != nullptr;
}
void fb_launch_gmm_wide(hipStream_t s, const FbGmmDev &g, const float *feats, const int *n_rows_ptr, int rows_cap,
                        int n_chunks, float *part_m, float *part_s) {
  // one launch per pass (fb_load_gmm: more than FB_FXW_MAX_M models are dealt over up to FB_FXW_MAX_PASS launches, each
  // with the base model and its own delta images; the base model's partials are written by every pass with the same
  // values)
  for (int p = 0; p < g.n_pass; ++p) {
    FbGmmDev gp = g;
    gp.images_fd = g.pass_images[p];
    gp.pass_first = g.pass_lo[p];
    switch (1 + g.pass_lo[p + 1] - g.pass_lo[p]) {
      case 2: launch_gmm_fxw_t<5, 2>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 3: launch_gmm_fxw_t<5, 3>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 4: launch_gmm_fxw_t<5, 4>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 5: launch_gmm_fxw_t<5, 5>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 6: launch_gmm_fxw_t<5, 6>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 7: launch_gmm_fxw_t<5, 7>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 8: launch_gmm_fxw_t<5, 8>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 9: launch_gmm_fxw_t<5, 9>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      case 10: launch_gmm_fxw_t<5, 10>(s, gp, feats, n_rows_ptr, rows_cap, n_chunks, part_m, part_s); break;
      default: break;  // fb_load_gmm deals at most FB_FXW_MAX_M - 1 delta models to a pass
    }
  }
}

