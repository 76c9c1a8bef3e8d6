// does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs?  (build: hipcc --offload-arch=gfx950 -O2 -o f16_denorm_probe ...)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float aval, float bval) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
  a[0] = (_Float16)aval;   // k = 0 (lanes 0..31) / k = 8 (lanes 32..63)
  b[0] = (_Float16)bval;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.0f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float *d; hipMalloc(&d, 4);
  const float as[] = {1.0f, 6.103515625e-05f, 3.0517578125e-05f, 9.5367431640625e-07f, 5.9604644775390625e-08f};
  for (float a : as) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, 1024.0f);
    float h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a=%g (f16 %s)  a*1024*2 = %g   expected %g\n", a, a < 6.1e-5f ? "subnormal" : "normal", h, 2.0f * a * 1024.0f);
  }
  return 0;
}
