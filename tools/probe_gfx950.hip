// Hardware-fact probe for gfx950: validates the lane maps this repo's kernels rely on
// (MFMA 16x16x32 / 32x32x16 bf16 operand + accumulator layouts, global_load_lds
// lane-linear destination with per-lane source, ds_read_b64_tr_b16 transpose read).
// Build: hipcc --offload-arch=gfx950 -O2 probe_gfx950.hip -o probe_gfx950 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

// A [16][32] row-major bf16, B [32][16] row-major (k-major), C [16][16]
__global__ void mfma16(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    unsigned short av = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    unsigned short bv = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    a[j] = __builtin_bit_cast(__bf16, av);
    b[j] = __builtin_bit_cast(__bf16, bv);
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
// A [32][16], B [16][32], C [32][32]
__global__ void mfma32(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = __builtin_bit_cast(__bf16, A[(l & 31) * 16 + 8 * (l >> 5) + j]);
    b[j] = __builtin_bit_cast(__bf16, B[(8 * (l >> 5) + j) * 32 + (l & 31)]);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// glds: lane l loads 16 B from src chunk perm[l]; LDS must then hold chunk perm[l] at byte 16*l (+ base 1024*wave)
__global__ void glds(const unsigned* src, const int* perm, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned smem[2 * 256];
  int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned* g = src + perm[threadIdx.x] * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
      (__attribute__((address_space(3))) void*)(smem + w * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = smem[w * 256 + l * 4 + i];
}
// tr read: LDS tile [R=16 rows][C=16 cols] of u16 with row stride 'stride' elements, value = row*100+col.
// Each 16-lane group g reads rows 4g..4g+3 (block), cols 0..15; lane 4q+p gives address of row q, cols 4p..4p+3.
__global__ void trread(short* out) {
  __shared__ __attribute__((aligned(16))) short t[16 * 16];
  int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) t[i] = (short)((i / 16) * 100 + (i % 16));
  __syncthreads();
  int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(t + (4 * g + q) * 16 + 4 * p));
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}

int main() {
  int fails = 0;
  {  // MFMA 16x16x32
    std::vector<unsigned short> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16), C(256), Cr(256, 0.f);
    for (int i = 0; i < 512; ++i) { Af[i] = (float)((i * 7 + 3) % 11 - 5); A[i] = f2bf(Af[i]); Bf[i] = (float)((i * 5 + 1) % 13 - 6); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) Cr[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
    unsigned short *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) if (C[i] != Cr[i]) ++bad;
    printf("mfma_f32_16x16x32_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // MFMA 32x32x16
    std::vector<unsigned short> A(512), B(512); std::vector<float> Af(512), Bf(512), C(1024), Cr(1024, 0.f);
    for (int i = 0; i < 512; ++i) { Af[i] = (float)((i * 7 + 3) % 11 - 5); A[i] = f2bf(Af[i]); Bf[i] = (float)((i * 5 + 1) % 13 - 6); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) Cr[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
    unsigned short *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) if (C[i] != Cr[i]) ++bad;
    printf("mfma_f32_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // glds
    std::vector<unsigned> src(128 * 4); std::vector<int> perm(128); std::vector<unsigned> out(128 * 4);
    for (int i = 0; i < 512; ++i) src[i] = 1000 + i;
    for (int i = 0; i < 128; ++i) perm[i] = (i * 37 + 11) % 128;
    unsigned *dS, *dO; int* dP; CK(hipMalloc(&dS, 2048)); CK(hipMalloc(&dO, 2048)); CK(hipMalloc(&dP, 512));
    CK(hipMemcpy(dS, src.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, perm.data(), 512, hipMemcpyHostToDevice));
    glds<<<1, 128>>>(dS, dP, dO); CK(hipMemcpy(out.data(), dO, 2048, hipMemcpyDeviceToHost));
    int bad = 0; for (int t = 0; t < 128; ++t) for (int i = 0; i < 4; ++i) if (out[t * 4 + i] != src[perm[t] * 4 + i]) ++bad;
    printf("global_load_lds x16 (per-lane src, lane-linear dst): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // tr read
    std::vector<short> out(256); short* dO; CK(hipMalloc(&dO, 512));
    trread<<<1, 64>>>(dO); CK(hipMemcpy(out.data(), dO, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; for (int e = 0; e < 4; ++e) { int exp = (4 * g + e) * 100 + i; if (out[l * 4 + e] != exp) ++bad; } }
    printf("ds_read_b64_tr_b16 (lane i gets col i, rows 0..3 of its group's block): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    if (bad) { for (int l = 0; l < 64; ++l) printf("lane %d: %d %d %d %d\n", l, out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]); }
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s CUs=%d clock=%d kHz lds/block=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  printf(fails ? "PROBE: %d FAILED\n" : "PROBE: ALL PASS\n", fails);
  return fails;
}
