// Shared between the convolution kernels (conv_igemm.hip: flat-M implicit GEMM; conv_halo.hip: halo-reuse 3x3).
#pragma once
#include "common.h"

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  const bf16_t* mask;
  const unsigned char* mask_bits;   // data gradient: the ReLU mask as one bit per element, [M][Co/8] bytes (danhip_relu_bits), instead of `mask`
  const bf16_t* resid;
  void* y;
  unsigned char* bits_out;        // forward conv_relu on the 128-wide halo tiles: also writes the ReLU bit mask of y ([M][Co/8] bytes) ...
  unsigned char* pool_bits_out;   // ... and of the fused pooled output
  bf16_t* pool_y;   // optional: 2x2/stride-2 SAME max-pool of y (ReLU outputs), written by the kernels that can fuse it
  unsigned char* pool_arg_out;    // optional, with pool_y: 2-bit arg-max codes of the pooled map ([pooled pixel][Co/4] bytes, channel c in bits
                                  // 2(c%4).. of byte c/4; code = 2*dh + dw of the FIRST maximum in row-major window order) for danhip_maxpool2x2_bwd_arg
  int N, H, W, C;
  int Ho, Wo, Co;
  int kh, kw, stride, pad_t, pad_l;
  int M, Kpad, ktiles, taps, cpt;
  FastDiv div_wo, div_howo, div_c, div_kw;
  int relu, out_f32, accumulate;
  int dshift;    // log2(dstride)
  float* splitk_ws;      // split-K (conv_igemm.hip, maps too small to fill the chip): fp32 partial outputs [splits][M][Co]; null = no split
  int splits, kt_per_split;
  int dstride;   // >1: strided data gradient — a source tap exists only where (h,w) are multiples of dstride (power of two)
  // Channel-slice views (round 4: the DAN context block writes its branches straight into the concat buffer and reads branch inputs
  // out of a wider tensor): pixel pitches in ELEMENTS of x, y and of the mask tensor of a data gradient (a slice [.., c0:c0+C] of an NHWC
  // tensor of width ld is its base pointer + c0 and pitch ld).  fwd_args / bwd_args set them to the dense values C / Co / Co;
  // conv_pointwise.hip, conv_igemm.hip (+ split-K finish) and conv_halo_c64.hip honour other values - strided() keeps the call off the rest.
  int ldx, ldy, ldm;
  int relu_co;   // forward: ReLU applies to output channels < relu_co only (fused 1x1 block whose last columns stay linear); default Co
  bool strided() const { return ldx != C || ldy != Co || (mask && ldm != Co) || (relu && relu_co < Co); }
  // (appended last: the layout of everything above is what the tuned kernels were built against)
  const bf16_t* x2;   // conv_pointwise.hip, forward 1x1: second source of the K axis (channels ksplit*64 .. C-1 of the virtual input [x | x2], same
  int ksplit;         // pixel pitch ldx as x); null = one source.  Only danhip_conv2d_fwd_concat2 sets them.
  // Split-operand evaluation (csrc/split_infer.hip; fp16 build, forward, Co % 8 == 0): y is the 3-limb-layout map [M][3 Co] of the NEXT
  // convolution — hi = half(v) at column co, lo = half(v - hi) at Co + co, hi again at 2 Co + co — written from the fp32 epilogue value.
  // Honoured by conv_halo.hip's general epilogue and by the flat-M kernel / split-K finish (conv_store4); every other kernel declines.
  int split_out;
  // conv_halo_c64.hip, data gradient of the SECOND layer (conv1_2) with the FIRST layer's weight gradient folded in (round 6): dx of this
  // call is conv1_1's dY; conv1_1 has no data gradient of its own, so dx's only reader would be conv_wgrad_c8.hip.  With fuse_dw set the
  // kernel keeps each dx tile in LDS, multiplies it with the tile's patch of the 8-channel image x8 (transposing reads, M = 12 tap slots x 4
  // channels, N = 64) into per-wave accumulators and never stores dx (y == NULL allowed): dw8 [3,3,cin_real,64] / db8 [64] fp32, atomically added.
  const bf16_t* fuse_x8;
  float* fuse_dw;
  float* fuse_db;
  int fuse_cin_real;
};

// hi / lo limbs of 4 fp32 values as two packed pairs each (the build's 16-bit type: IEEE half where split_out is allowed)
__device__ __forceinline__ void dh_split4(const float v[4], uint2& hi, uint2& lo) {
  hi.x = pack2bf(v[0], v[1]);
  hi.y = pack2bf(v[2], v[3]);
  float h[4];
  unpack2bf(hi.x, h[0], h[1]);
  unpack2bf(hi.y, h[2], h[3]);
  lo.x = pack2bf(v[0] - h[0], v[1] - h[1]);
  lo.y = pack2bf(v[2] - h[2], v[3] - h[3]);
}

// true when launch_conv's kernel for these args writes a.pool_y itself (conv_halo_c64.hip / conv_halo.hip forward tiles)
bool danhip_conv_pool_fusable(const ConvArgs& a);
bool danhip_conv_halo_pool_fusable(const ConvArgs& a);
bool danhip_conv_c64_eligible(const ConvArgs& a);

// packed max of two pairs of NON-NEGATIVE 16-bit floats (ReLU outputs): they order like signed 16-bit integers, and a
// stray -0.0 (0x8000) is the smallest value.  One v_pk_max_i16.
typedef __attribute__((ext_vector_type(2))) short s16x2;
__device__ __forceinline__ unsigned pkmax_relu(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

// ---- epilogue arithmetic shared by the tile kernels (measured with tools/halo2_trace.hip: the epilogues, not the waits around them, were
// the longest stretch without MFMAs).
// ReLU in ONE instruction (fmaxf compiles to a canonicalising v_max_f32 plus the max); max(NaN, 0) = 0 like fmaxf.
__device__ __forceinline__ float dh_relu(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}
// Two packed 16-bit ReLU OUTPUTS (>= +0 and never NaN after dh_relu) -> 1 per half that is > 0: min(x, 1) as unsigned 16-bit integers
// (clang lowers the generic vector min to compares and selects, hence the asm).
typedef __attribute__((ext_vector_type(2))) unsigned short dh_u16x2;
typedef __attribute__((ext_vector_type(2))) unsigned dh_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned dh_u32x4;
__device__ __forceinline__ dh_u16x2 dh_pos2(unsigned packed) {
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(packed), "s"(0x00010001u));
  return __builtin_bit_cast(dh_u16x2, r);
}
// bits | (value r of the 8 packed ReLU outputs > 0) << (SHIFT + r): one v_pk_min_u16 + one v_dot2_u32_u16 per register (danhip_relu_bits
// layout: bit r of the byte = channel 8 k + r)
template <int SHIFT>
__device__ __forceinline__ unsigned dh_pos_bits8_acc(const dh_u32x4& t, unsigned bits) {
  static_assert(SHIFT == 0 || SHIFT == 8, "16-bit weights");
#pragma unroll
  for (int e = 0; e < 4; ++e)
    bits = __builtin_amdgcn_udot2(dh_pos2(t[e]), dh_u16x2{(unsigned short)(1u << (SHIFT + 2 * e)), (unsigned short)(2u << (SHIFT + 2 * e))}, bits, false);
  return bits;
}
// ---- 2-bit arg-max codes of a 2x2 max-pool window, from the packed ReLU outputs the pooling epilogues hold (round 4: the pool's backward
// then scatters through the codes instead of re-reading the full-resolution activation).  a / b / c = top-left / top-right / bottom-left
// values of two channels per 32-bit word, m = the window maximum (pkmax_relu); all non-negative 16-bit floats, so they order as unsigned
// integers and m - x is 0 exactly where x is a maximum (packed ops: no borrow between the halves).  TF's MaxPoolGrad takes the FIRST maximum
// in row-major order (the rule of maxpool_bwd_kernel): code = 0 if a == m, else 1 if b == m, else 2 if c == m, else 3 = ne_a (1 + ne_b (1 + ne_c)).
__device__ __forceinline__ unsigned dh_pk_ne(unsigned m, unsigned x) {      // per half: 0 where x == m, 1 elsewhere
  unsigned d, r;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(m), "v"(x));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(d), "s"(0x00010001u));
  return r;
}
__device__ __forceinline__ unsigned dh_argmax2x2_pk(unsigned a, unsigned b, unsigned c, unsigned m) {   // -> per half a code 0..3
  const unsigned na = dh_pk_ne(m, a), nb = dh_pk_ne(m, b), nc = dh_pk_ne(m, c);
  unsigned t, r;
  asm("v_pk_mad_u16 %0, %1, %2, %1" : "=v"(t) : "v"(nb), "v"(nc));         // nb * nc + nb
  asm("v_pk_mad_u16 %0, %1, %2, %1" : "=v"(r) : "v"(na), "v"(t));          // na * t + na
  return r;
}
// the eight codes of a lane's eight consecutive channels (four words of two) as 16 bits: channel j in bits 2j, 2j+1
__device__ __forceinline__ unsigned dh_argmax2x2_codes16(const dh_u32x4& a, const dh_u32x4& b, const dh_u32x4& c, const dh_u32x4& m) {
  unsigned out = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned w = dh_argmax2x2_pk(a[e], b[e], c[e], m[e]);
    out |= ((w | (w >> 14)) & 0xFu) << (4 * e);
  }
  return out;
}

// x | x(lane ^ 16) | x(lane ^ 32) | x(lane ^ 48) with the gfx950 row swaps (VALU) instead of two ds_bpermute round trips:
//   v_permlane16_swap(x, x) -> {rows [0,0,2,2], rows [1,1,3,3]};  v_permlane32_swap(y, y) -> {lower half twice, upper half twice}
__device__ __forceinline__ unsigned dh_or_rows(unsigned x) {
  const dh_u32x2 a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const unsigned y = a[0] | a[1];
  const dh_u32x2 b = __builtin_amdgcn_permlane32_swap(y, y, false, false);
  return b[0] | b[1];
}
// value of lane ^ 1 (DPP quad_perm [1,0,3,2])
__device__ __forceinline__ unsigned dh_lane_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }

// Halo-reuse 3x3/stride-1 kernel (conv_halo.hip).  Returns DANHIP_OK when it launched, 1 when the shape is not
// eligible (caller falls back to the flat-M kernel), negative on a launch error.
int danhip_launch_conv_halo(const ConvArgs& a, hipStream_t s);
bool danhip_conv_halo_takes_bits(const ConvArgs& a);
bool danhip_conv_halo_emits_bits(const ConvArgs& a);   // the forward instance for these args writes ConvArgs::bits_out (and pool_bits_out with a fused pool)   // the data-gradient instance for these args reads ConvArgs::mask_bits
const char* danhip_conv_halo_label(const ConvArgs& a, bool dgrad);
// 64 -> 64 channel special case with register-resident weights (conv_halo_c64.hip); same return convention.
int danhip_launch_conv_c64(const ConvArgs& a, hipStream_t s);
const char* danhip_conv_c64_label(const ConvArgs& a, bool dgrad);   // kernel-instance label or nullptr when not eligible

// 8 (= 3 padded) -> 64 channel first layer, store-bound (conv_c8.hip); same return convention, forward only.
int danhip_launch_conv_c8(const ConvArgs& a, hipStream_t s);
const char* danhip_conv_c8_label(const ConvArgs& a);

// Pointwise (1x1 / stride 1) streaming GEMM (conv_pointwise.hip): forward (bias, ReLU) and data gradient (mask, accumulate); same return convention.
int danhip_launch_conv_pointwise(const ConvArgs& a, hipStream_t s);
const char* danhip_conv_pointwise_label(const ConvArgs& a, bool dgrad);

// Row-streaming 3x3/stride-1 weight gradient with a register window of X fragments (conv_wgrad_rows.hip): DANHIP_OK when launched,
// 1 when the shape is not eligible.
const char* danhip_wgrad_rows_label(const danhip_conv_desc* d);
// Pointwise (1x1 / stride 1) weight gradient, 256 x 256 gradient tile per workgroup (conv_wgrad_pw.hip); same return convention.
const char* danhip_wgrad_pw_label(const danhip_conv_desc* d);
int danhip_launch_wgrad_pw(const danhip_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* db, int cin_real, hipStream_t s,
                           void* ws = nullptr, size_t ws_bytes = 0, int ldx = 0, int ldy = 0);      // ldx / ldy: pixel pitches (0 = dense)
size_t danhip_wgrad_pw_workspace_bytes(const danhip_conv_desc* d);
int danhip_launch_wgrad_rows(const danhip_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* db, int cin_real, hipStream_t s,
                             void* ws = nullptr, size_t ws_bytes = 0, int ldx = 0, int ldy = 0);      // ldx / ldy: pixel pitches (0 = dense)
size_t danhip_wgrad_rows_workspace_bytes(const danhip_conv_desc* d);
// First-layer weight gradient (3x3 / stride 1, 8-channel image with <= 4 real channels, 64 outputs; conv_wgrad_c8.hip)
bool danhip_wgrad_c8_eligible(const danhip_conv_desc* d, int cin_real, int ldx, int ldy);
int danhip_launch_wgrad_c8(const danhip_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* db, int cin_real, hipStream_t s);

// Epilogue for one lane's 4 consecutive output channels [co, co+4) of output pixel m (shared by both kernels).
__device__ __forceinline__ void conv_store4(const ConvArgs& a, float v[4], size_t m, int co) {
  if (co >= a.Co) return;
  const size_t o = m * (size_t)(a.split_out ? 3 * a.ldy : a.ldy) + co;      // output (pitched view: ldy elements per pixel; limb layout: three of them)
  const size_t om = m * (size_t)a.ldm + co;         // mask tensor of a data gradient
  const size_t orr = m * (size_t)a.Co + co;         // residual: always dense
  const bool full = (co + 4 <= a.Co) && ((a.Co & 3) == 0);
  if (a.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (co + r < a.Co) v[r] += a.bias[co + r];
  }
  if (a.relu && co < a.relu_co) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (a.out_f32) {
    float* y = reinterpret_cast<float*>(a.y) + o;
    if (full) {
      float4 t = make_float4(v[0], v[1], v[2], v[3]);
      if (a.accumulate) { float4 u = *reinterpret_cast<float4*>(y); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      *reinterpret_cast<float4*>(y) = t;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (co + r < a.Co) y[r] = a.accumulate ? y[r] + v[r] : v[r];
    }
  } else {
    bf16_t* y = reinterpret_cast<bf16_t*>(a.y) + o;
    if (a.split_out) {                                  // (forward only, Co % 8 == 0: always a full quad)
      uint2 hi, lo;
      dh_split4(v, hi, lo);
      *reinterpret_cast<uint2*>(y) = hi;
      *reinterpret_cast<uint2*>(y + a.Co) = lo;
      *reinterpret_cast<uint2*>(y + 2 * a.Co) = hi;
      return;
    }
    if (full) {
      if (a.mask) {
        const uint2 mk = *reinterpret_cast<const uint2*>(a.mask + om);
        const bf16_t* mp = reinterpret_cast<const bf16_t*>(&mk);
#pragma unroll
        for (int r = 0; r < 4; ++r) if (!(bf2f(mp[r]) > 0.f)) v[r] = 0.f;
      }
      if (a.resid) {
        const uint2 rs = *reinterpret_cast<const uint2*>(a.resid + orr);
        const bf16_t* rp = reinterpret_cast<const bf16_t*>(&rs);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f(rp[r]);
      }
      if (a.accumulate) {
        const uint2 old = *reinterpret_cast<const uint2*>(y);
        const bf16_t* op = reinterpret_cast<const bf16_t*>(&old);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f(op[r]);
      }
      uint2 t;
      t.x = pack2bf(v[0], v[1]);
      t.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(y) = t;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (co + r < a.Co) {
          float t = v[r];
          if (a.mask && !(bf2f(a.mask[om + r]) > 0.f)) t = 0.f;
          if (a.resid) t += bf2f(a.resid[orr + r]);
          if (a.accumulate) t += bf2f(y[r]);
          y[r] = f2bf(t);
        }
      }
    }
  }
}
