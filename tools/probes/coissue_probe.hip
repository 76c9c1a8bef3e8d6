// co-issue probe for the k_gmm_fx2 redesign: how many logsumexp VALU instructions hide behind f16 MFMAs
//   MODE 0: 30 MFMAs on ONE accumulator, back to back, then the epilogue of both result sets (today's structure x2)
//   MODE 1: 30 MFMAs alternating TWO accumulators, then the epilogue
//   MODE 2: alternating accumulators with the PREVIOUS phase's epilogue interleaved (sched_group_barrier), unpacked f32
//   MODE 3: as 2 but packed f32 (v_pk_fma / v_pk_add)
//   MODE 4: MFMAs only (alternating), no epilogue
//   MODE 5: epilogue only
// hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MF(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)

__device__ __forceinline__ void lse16(const f32x16 &pv, float &m, float &s, float ls) {
  float tm = pv[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) tm = fmaxf(tm, pv[r]);
  const float mn = fmaxf(m, tm);
  const float rn = mn * ls, ro = m * ls;
  float acc = s * __builtin_amdgcn_exp2f(ro - rn);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc += __builtin_amdgcn_exp2f(__builtin_fmaf(pv[r], ls, -rn));
  m = mn; s = acc;
}
__device__ __forceinline__ void lse16_pk(const f32x16 &pv, float &m, float &s, float ls) {
  float tm = pv[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) tm = fmaxf(tm, pv[r]);
  const float mn = fmaxf(m, tm);
  const float rn = mn * ls, ro = m * ls;
  const f32x2 l2 = {ls, ls}, nr2 = {-rn, -rn};
  f32x2 acc = {s * __builtin_amdgcn_exp2f(ro - rn), 0.0f};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const f32x2 v2 = {pv[2 * r], pv[2 * r + 1]};
    const f32x2 t = __builtin_elementwise_fma(v2, l2, nr2);
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    acc += e;
  }
  m = mn; s = acc[0] + acc[1];
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, long long *cyc, int iters, float ls, const u32x4 *gsrc = nullptr) {
  const int lane = threadIdx.x & 63;
  extern __shared__ __attribute__((aligned(16))) u32x4 sm[];  // 4 slots x 10 KB used; 100 KB requested so that ONE workgroup fits a CU
  if (MODE >= 13) { for (int i = threadIdx.x; i < 4 * 640; i += 256) sm[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x34003400u}; __syncthreads(); }
  f16x8 a[5], b[5];
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[c][i] = (_Float16)(0.01f * ((lane + i + c) & 15)); b[c][i] = (_Float16)(0.02f * ((lane * 3 + i + c) & 7)); }
  f16x8 a2[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) a2[c] = a[(c + 1) % 5];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned sm_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)sm;
  f32x16 accA, accB, pvA, pvB;
#pragma unroll
  for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; pvA[r] = -1.0f * r; pvB[r] = -2.0f * r; }
  float m0 = -1e30f, s0 = 0.f, m1 = -1e30f, s1 = 0.f;
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 5; ++c) { MF(a[c], b[c], accA); MF(a[c], b[(c + 1) % 5], accA); MF(a[(c + 2) % 5], b[c], accA); }
#pragma unroll
      for (int c = 0; c < 5; ++c) { MF(a[c], b[(c + 3) % 5], accB); MF(a[(c + 1) % 5], b[c], accB); MF(a[(c + 4) % 5], b[c], accB); }
      lse16_pk(accA, m0, s0, ls); lse16_pk(accB, m1, s1, ls);
    } else if (MODE == 1) {
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        MF(a[c], b[c], accA); MF(a[c], b[(c + 3) % 5], accB); MF(a[c], b[(c + 1) % 5], accA);
        MF(a[(c + 1) % 5], b[c], accB); MF(a[(c + 2) % 5], b[c], accA); MF(a[(c + 4) % 5], b[c], accB);
      }
      lse16_pk(accA, m0, s0, ls); lse16_pk(accB, m1, s1, ls);
    } else if (MODE == 2 || MODE == 3) {
      // two phases per trip with explicit ping-pong register sets: MFMAs write (accA, accB) while the epilogue
      // consumes (pvA, pvB), then the roles swap
#define PHASE(XA, XB, YA, YB)                                                                                  \
      {                                                                                                        \
        if (MODE == 2) { lse16(YA, m0, s0, ls); lse16(YB, m1, s1, ls); } else { lse16_pk(YA, m0, s0, ls); lse16_pk(YB, m1, s1, ls); } \
        _Pragma("unroll") for (int c = 0; c < 5; ++c) {                                                         \
          if (c == 0) { XA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[c], zero, 0, 0, 0);                 \
                        XB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[(c + 3) % 5], zero, 0, 0, 0); }     \
          else { MF(a[c], b[c], XA); MF(a[c], b[(c + 3) % 5], XB); }                                            \
          MF(a[c], b[(c + 1) % 5], XA); MF(a[(c + 1) % 5], b[c], XB); MF(a[(c + 2) % 5], b[c], XA); MF(a[(c + 4) % 5], b[c], XB); \
        }                                                                                                      \
        _Pragma("unroll") for (int g = 0; g < 30; ++g) {                                                        \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                    \
          __builtin_amdgcn_sched_group_barrier(0x002, MODE == 2 ? 4 : 3, 0);                                    \
        }                                                                                                      \
      }
      PHASE(accA, accB, pvA, pvB)
      PHASE(pvA, pvB, accA, accB)
      ++it;
    } else if (MODE == 6 || MODE == 7 || MODE == 8) {
      // hand-placed: every gap = 1 MFMA (alternating accumulators) + NV value updates (fma, exp, add) + one max3;
      // MODE 6: 1 value per gap (3+1 VALU), MODE 7: 2 values per gap (6+1 VALU), MODE 8: MFMA + 1 max3 only
#define GAP(ACC, AF, BF, V0, V1, MX0, MX1)                                                                    \
      if (MODE == 6)                                                                                          \
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_fma_f32 %1, %6, %8, %9\n v_max3_f32 %3, %3, %10, %11\n v_exp_f32 %1, %1\n s_nop 0\n v_add_f32 %2, %2, %1" \
                     : "+v"(ACC), "=&v"(t0), "+v"(sum0), "+v"(mx) : "v"(AF), "v"(BF), "v"(V0), "v"(V1), "v"(ls), "v"(nr), "v"(MX0), "v"(MX1)); \
      else if (MODE == 7)                                                                                     \
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %5, %6, %0\n v_fma_f32 %1, %7, %9, %10\n v_fma_f32 %2, %8, %9, %10\n v_max3_f32 %4, %4, %11, %12\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %1\n v_add_f32 %3, %3, %2" \
                     : "+v"(ACC), "=&v"(t0), "=&v"(t1), "+v"(sum0), "+v"(mx) : "v"(AF), "v"(BF), "v"(V0), "v"(V1), "v"(ls), "v"(nr), "v"(MX0), "v"(MX1)); \
      else                                                                                                    \
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_max3_f32 %1, %1, %4, %5" : "+v"(ACC), "+v"(mx) : "v"(AF), "v"(BF), "v"(MX0), "v"(MX1));
      float t0, t1, nr = -m0 * ls, sum0 = s0, mx = m1;
#define PHASE6(XA, XB, YA, YB)                                                                                 \
      _Pragma("unroll") for (int c = 0; c < 5; ++c) {                                                           \
        GAP(XA, a[c], b[c], YA[(6 * c + 0) & 15], YB[(6 * c + 0) & 15], XA[0], XA[1])                            \
        GAP(XB, a[c], b[(c + 3) % 5], YA[(6 * c + 1) & 15], YB[(6 * c + 1) & 15], XA[2], XA[3])                  \
        GAP(XA, a[c], b[(c + 1) % 5], YA[(6 * c + 2) & 15], YB[(6 * c + 2) & 15], XA[4], XA[5])                  \
        GAP(XB, a[(c + 1) % 5], b[c], YA[(6 * c + 3) & 15], YB[(6 * c + 3) & 15], XB[0], XB[1])                  \
        GAP(XA, a[(c + 2) % 5], b[c], YA[(6 * c + 4) & 15], YB[(6 * c + 4) & 15], XB[2], XB[3])                  \
        GAP(XB, a[(c + 4) % 5], b[c], YA[(6 * c + 5) & 15], YB[(6 * c + 5) & 15], XB[4], XB[5])                  \
      }
      PHASE6(accA, accB, pvA, pvB)
      PHASE6(pvA, pvB, accA, accB)
      s0 = sum0; m1 = mx;
      ++it;
    } else if (MODE == 9 || MODE == 10 || MODE == 11) {
      // 9: 30 alternating MFMAs strictly back to back, THEN (sched_barrier) the previous phase's two epilogues
      // 10: the epilogues FIRST, then the 30 MFMAs
#define PHASE9(XA, XB, YA, YB)                                                                                 \
      {                                                                                                        \
        if (MODE == 10) { lse16_pk(YA, m0, s0, ls); lse16_pk(YB, m1, s1, ls); __builtin_amdgcn_sched_barrier(0); } \
        _Pragma("unroll") for (int c = 0; c < 5; ++c) {                                                         \
          if (c == 0) { XA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[c], zero, 0, 0, 0);                 \
                        XB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[(c + 3) % 5], zero, 0, 0, 0); }     \
          else { MF(a[c], b[c], XA); MF(a[c], b[(c + 3) % 5], XB); }                                            \
          MF(a[c], b[(c + 1) % 5], XA); MF(a[(c + 1) % 5], b[c], XB); MF(a[(c + 2) % 5], b[c], XA); MF(a[(c + 4) % 5], b[c], XB); \
        }                                                                                                      \
        if (MODE == 11) asm volatile("" : "+a"(XA), "+a"(XB));  /* accumulators live in AGPRs */              \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (MODE == 9 || MODE == 11) { lse16_pk(YA, m0, s0, ls); lse16_pk(YB, m1, s1, ls); __builtin_amdgcn_sched_barrier(0); } \
      }
      PHASE9(accA, accB, pvA, pvB)
      PHASE9(pvA, pvB, accA, accB)
      ++it;
    } else if (MODE >= 12 && MODE <= 15) {
      // 12: mode 11 (MFMAs, then previous epilogues; AGPR accumulators) + a barrier per phase
      // 13: + the A operands come from LDS (10 ds_read_b128 per phase, prefetched one phase ahead into a second set)
      // 14: + 3 LDS-DMA pieces per wave and phase (ring, counted vmcnt), like k_gmm_fx2w
      // 15: as 14 with the DMA issued BEFORE the MFMAs
#define PHASE12(XA, XB, YA, YB, FA, FB, PH)                                                                    \
      {                                                                                                        \
        if (MODE == 15) { for (int u = 0; u < 3; ++u) glds16(gsrc + (size_t)(((it + PH) * 7 + u) & 1023) * 64 + lane, sm_lds + (unsigned)((((it + PH) & 3) * 640 + (wv + 4 * u) * 64) * 16)); } \
        _Pragma("unroll") for (int c = 0; c < 5; ++c) {                                                         \
          if (c == 0) { XA = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[c], b[c], zero, 0, 0, 0);                \
                        XB = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[c], b[(c + 3) % 5], zero, 0, 0, 0); }    \
          else { MF(FA[c], b[c], XA); MF(FA[c], b[(c + 3) % 5], XB); }                                          \
          MF(FA[c], b[(c + 1) % 5], XA); MF(FA[(c + 1) % 5], b[c], XB); MF(FA[(c + 2) % 5], b[c], XA); MF(FA[(c + 4) % 5], b[c], XB); \
        }                                                                                                      \
        asm volatile("" : "+a"(XA), "+a"(XB));                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (MODE == 14) { for (int u = 0; u < 3; ++u) glds16(gsrc + (size_t)(((it + PH) * 7 + u) & 1023) * 64 + lane, sm_lds + (unsigned)((((it + PH) & 3) * 640 + (wv + 4 * u) * 64) * 16)); } \
        if (MODE >= 13) { _Pragma("unroll") for (int c = 0; c < 5; ++c) FB[c] = __builtin_bit_cast(f16x8, sm[((it + PH + 1) & 3) * 640 + c * 64 + lane]); } \
        lse16_pk(YA, m0, s0, ls); lse16_pk(YB, m1, s1, ls);                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (MODE >= 14) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                       \
        __syncthreads();                                                                                       \
      }
      PHASE12(accA, accB, pvA, pvB, a, a2, 0)
      PHASE12(pvA, pvB, accA, accB, a2, a, 1)
      ++it;
    } else if (MODE == 16 || MODE == 17) {
      // 16: mode 9 with asm MFMAs -- accumulators in VGPRs, B operands read from AGPRs -- + s_nop 10 behind the block
      // 17: the same block with builtin MFMAs (reference for the checksum)
#define AMF(A, B, C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C) : "v"(A), "a"(B))
#define AMF0(A, B, C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(C) : "v"(A), "a"(B))
#define PHASE16(XA, XB, YA, YB)                                                                                \
      {                                                                                                        \
        _Pragma("unroll") for (int c = 0; c < 5; ++c) {                                                         \
          if (MODE == 16) {                                                                                    \
            if (c == 0) { AMF0(a[c], b[c], XA); AMF0(a[c], b[(c + 3) % 5], XB); }                               \
            else { AMF(a[c], b[c], XA); AMF(a[c], b[(c + 3) % 5], XB); }                                        \
            AMF(a[c], b[(c + 1) % 5], XA); AMF(a[(c + 1) % 5], b[c], XB); AMF(a[(c + 2) % 5], b[c], XA); AMF(a[(c + 4) % 5], b[c], XB); \
          } else {                                                                                             \
            if (c == 0) { XA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[c], zero, 0, 0, 0);               \
                          XB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[(c + 3) % 5], zero, 0, 0, 0); }   \
            else { MF(a[c], b[c], XA); MF(a[c], b[(c + 3) % 5], XB); }                                          \
            MF(a[c], b[(c + 1) % 5], XA); MF(a[(c + 1) % 5], b[c], XB); MF(a[(c + 2) % 5], b[c], XA); MF(a[(c + 4) % 5], b[c], XB); \
          }                                                                                                    \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        lse16_pk(YA, m0, s0, ls); lse16_pk(YB, m1, s1, ls);                                                    \
        if (MODE == 16) asm volatile("s_nop 10" : "+v"(XA), "+v"(XB));                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
      }
      PHASE16(accA, accB, pvA, pvB)
      PHASE16(pvA, pvB, accA, accB)
      ++it;
    } else if (MODE == 4) {
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        MF(a[c], b[c], accA); MF(a[c], b[(c + 3) % 5], accB); MF(a[c], b[(c + 1) % 5], accA);
        MF(a[(c + 1) % 5], b[c], accB); MF(a[(c + 2) % 5], b[c], accA); MF(a[(c + 4) % 5], b[c], accB);
      }
    } else {
      lse16_pk(pvA, m0, s0, ls); lse16_pk(pvB, m1, s1, ls);
#pragma unroll
      for (int r = 0; r < 16; ++r) { pvA[r] += 0.001f; pvB[r] -= 0.001f; }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = m0 + s0 + m1 + s1;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += accA[r] + accB[r] + pvA[r] + pvB[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int iters) {
  float *out; long long *cyc;
  hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  static u32x4 *gsrc = nullptr;
  if (!gsrc) { hipMalloc(&gsrc, 1024 * 64 * 16); hipMemset(gsrc, 0x3c, 1024 * 64 * 16); }
  const size_t ldsb = blocks > 256 ? 60 * 1024 : 100 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  probe<MODE><<<blocks, 256, ldsb>>>(out, cyc, iters, 0.0056f, gsrc);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, 256, ldsb>>>(out, cyc, iters, 0.0056f, gsrc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  float ho[256]; hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  double cs = 0; for (int i = 0; i < 256; ++i) cs += ho[i];
  const double phases = (double)iters;
  printf("%-44s blocks %4d (%d waves/SIMD): %.3f ms, %lld cycles -> %.0f cycles/phase (30 MFMA + 2 epilogues); clock %.2f GHz; checksum %.6e\n", name, blocks,
         blocks * 4 / 1024, ms, h[0], h[0] / phases, h[0] / (ms * 1e6), cs);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int b : {256, 512}) {
    run<0>("0 one accumulator, then epilogue (pk)", b, 2000);
    run<1>("1 two accumulators, then epilogue (pk)", b, 2000);
    run<2>("2 two acc + interleaved prev epilogue (f32)", b, 2000);
    run<3>("3 two acc + interleaved prev epilogue (pk)", b, 2000);
    run<4>("4 MFMA only (two acc)", b, 2000);
    run<5>("5 epilogue only (pk)", b, 2000);
    run<6>("6 asm gaps: MFMA + 1 value (fma,exp,add)+max3", b, 2000);
    run<7>("7 asm gaps: MFMA + 2 values (6 VALU)+max3", b, 2000);
    run<8>("8 asm gaps: MFMA + max3 only", b, 2000);
    run<9>("9 30 MFMAs, then prev epilogues (no interleave)", b, 2000);
    run<10>("10 prev epilogues, then 30 MFMAs", b, 2000);
    run<11>("11 as 9 with the accumulators in AGPRs", b, 2000);
    run<12>("12 = 11 + barrier per phase", b, 2000);
    run<13>("13 = 12 + A operands prefetched from LDS", b, 2000);
    run<14>("14 = 13 + 3 LDS-DMA pieces per phase (after MFMAs)", b, 2000);
    run<15>("15 = 13 + 3 LDS-DMA pieces per phase (before MFMAs)", b, 2000);
    run<16>("16 = 9 with asm MFMAs (acc VGPR, B from AGPR)", b, 2000);
    run<17>("17 = 9 again (builtin; checksum reference)", b, 2000);
  }
  return 0;
}
