// Shared device/host helpers for libdanhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/danhip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16_t;  // raw bf16 storage

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

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

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-safe)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (two scalar casts cost cvt+cvt+shift+or)
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

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
