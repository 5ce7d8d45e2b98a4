// Shared device/host helpers for libdanhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/danhip.h"

// 16-bit activation storage type of this build: bf16 (libdanhip.so, default) or fp16 (libdanhip_f16.so, -DDANHIP_FP16;
// BASELINE.json configs[4]).  The historical names bf16_t / bf2f / f2bf / pack2bf / bf16x8 mean "the build's 16-bit type".
#ifdef DANHIP_FP16
typedef _Float16 act16_t;
#define DH_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define DH_POS_INF16 0x7c00u      /* bit pattern of +inf in the build's 16-bit type: patterns 1 .. +inf are the values > 0 */
#else
typedef __bf16 act16_t;
#define DH_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define DH_POS_INF16 0x7f80u
#endif
typedef __attribute__((ext_vector_type(8))) act16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) act16_t bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16_t;  // raw bf16 storage

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// Zero-fill as an ordinary kernel launch (elementwise.hip).  Used instead of hipMemsetAsync for multi-megabyte scratch: a
// 5.7 MB memset NODE recorded in a hipGraph did not take effect on replay (ROCm 7.2; found with rocgdb in
// route_train_cell_kernel reading unset bucket counts), kernel nodes do.  ptr 16-byte aligned, bytes a multiple of 4.
int danhip_zero_async(void* ptr, size_t bytes, hipStream_t stream);

// kernel-selection switch by name (capi.cpp: danhip_set_option / environment default)
int danhip_option(const char* name);

// ---------------------------------------------------------------- error plumbing (never throws across the ABI)
void danhip_set_error(const char* fmt, ...);
#define DH_REQUIRE(cond, code, ...)                  \
  do {                                               \
    if (!(cond)) {                                   \
      danhip_set_error(__VA_ARGS__);                 \
      return (code);                                 \
    }                                                \
  } while (0)
#define DH_LAUNCH_CHECK()                                                        \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      danhip_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return DANHIP_ELAUNCH;                                                     \
    }                                                                            \
  } while (0)

// ---------------------------------------------------------------- 16-bit activation <-> f32
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) act16_t bf16x2;
#ifdef DANHIP_FP16
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// (w & 0xffff) and (w >> 16) halves of a packed pair
__device__ __forceinline__ void unpack2bf(unsigned w, float& lo, float& hi) {
  const bf16x2 h = __builtin_bit_cast(bf16x2, w);
  lo = (float)h[0]; hi = (float)h[1];
}
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ void unpack2bf(unsigned w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
#endif
// acc + lo(a) * lo(b) + hi(a) * hi(b) on packed 16-bit pairs: v_dot2c_f32_bf16 / v_dot2c_f32_f16 (products exact in fp32, fp32 accumulate) -
// a dot product over packed activations without unpacking them first
__device__ __forceinline__ float dh_dot2(unsigned a, unsigned b, float acc) {
#ifdef DANHIP_FP16
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
#endif
}
__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even (bf16: v_cvt_pk_bf16_f32, NaN-safe; fp16: v_cvt_f16_f32)
  const act16_t b = (act16_t)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {  // one packed convert (two scalar casts cost cvt+cvt+shift+or)
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// ---------------------------------------------------------------- fast division by a runtime constant
struct FastDiv {
  unsigned d, mul, shr;
};
static inline FastDiv make_fastdiv(unsigned d) {  // valid for n < 2^31
  FastDiv f;
  f.d = d;
  if (d == 1) { f.mul = 0; f.shr = 0; return f; }
  unsigned l = 0;
  while ((1u << l) < d) ++l;  // ceil(log2 d)
  uint64_t m = ((uint64_t)1 << (32 + l - 1)) / d + 1;  // round-up magic for 31-bit numerators
  f.mul = (unsigned)m;
  f.shr = l - 1;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

// 16-byte aligned page of zeros used as the source of padding / out-of-range LDS-DMA lanes.
static __device__ __attribute__((aligned(64))) unsigned g_danhip_zero_page[64] = {0};  // one copy per translation unit

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_dst_wave_base, 16, 0, 0);
}

// 16-byte LDS-DMA through a buffer descriptor: address = base + voff (per lane) + soff (uniform); a lane whose voff is out of
// range (0xFFFFFFFF) writes ZEROS to its LDS slot.
__device__ __forceinline__ void bufdma16_lds(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, soff, 0, 0);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
