// Does v_mfma_f32_16x16x32_f16 keep fp16 subnormal INPUTS (item: split-operand evaluation, x = hi + lo with lo often subnormal)?
// A = all `a`, B = all `b`; C[0] = 32*a*b expected.  Prints the products for subnormal a (2^-20 .. 2^-24) against b = 1 and b = 1024.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_f16_denorm.hip -o tools/probe_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const _Float16* ab, float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = ab[0]; b[j] = ab[1]; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  _Float16* d; float* o;
  hipMalloc(&d, 4); hipMalloc(&o, 4);
  int bad = 0;
  for (int e = -13; e >= -24; --e) {
    for (float bv : {1.0f, 1024.0f, 6.1035e-5f * 0.5f}) {
      _Float16 h[2] = {(_Float16)ldexpf(1.5f, e), (_Float16)bv};
      hipMemcpy(d, h, 4, hipMemcpyHostToDevice);
      k<<<1, 64>>>(d, o);
      float got; hipMemcpy(&got, o, 4, hipMemcpyDeviceToHost);
      float want = 32.f * (float)h[0] * (float)h[1];
      printf("a=1.5*2^%d (%s) b=%g : got %.9g want %.9g %s\n", e, e < -14 ? "subnormal" : "normal", (double)(float)h[1], got, want, got == want ? "OK" : "MISMATCH");
      bad += got != want;
    }
  }
  printf("F16_DENORM_PROBE %s\n", bad ? "FLUSHES_OR_DIFFERS" : "SUBNORMALS_KEPT");
  return 0;
}
