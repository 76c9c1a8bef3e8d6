// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// (k_gmm_fx2w brings four consecutive 1 KB pieces with one M0 set-up if it does.)
// hipcc --offload-arch=gfx950 -O3 -o glds_offset_probe glds_offset_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4 *src, u32x4 *dst) {
  __shared__ __attribute__((aligned(16))) u32x4 sm[512];  // 8 KB
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) sm[i] = u32x4{0xdeadbeefu, 0, 0, 0};
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)sm;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %1, off offset:1024\n\t"
               "global_load_lds_dwordx4 %1, off offset:2048\n\t"
               "global_load_lds_dwordx4 %1, off offset:3072\n\t"
               "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(src + lane), "s"(base) : "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) dst[i] = sm[i];
}
int main() {
  u32x4 *src, *dst; hipMalloc(&src, 8192); hipMalloc(&dst, 8192);
  unsigned h[2048]; for (int i = 0; i < 2048; ++i) h[i] = i;
  hipMemcpy(src, h, 8192, hipMemcpyHostToDevice);
  k<<<1, 64>>>(src, dst); hipDeviceSynchronize();
  unsigned o[2048]; hipMemcpy(o, dst, 8192, hipMemcpyDeviceToHost);
  int ok = 1; for (int i = 0; i < 1024; ++i) if (o[i] != (unsigned)i) { ok = 0; printf("first mismatch at word %d: %08x\n", i, o[i]); break; }
  printf("LDS follows the instruction offset: %s; word 1024 (untouched?) = %08x\n", ok ? "YES" : "NO", o[1024]);
  return 0;
}
