// SPLIT-OPERAND inference: fp32-accurate evaluation at the 16-bit MFMA rate (north_star: "eval_dan.py box outputs
// within 1e-4 of the reference on identical weights/inputs", /root/reference/eval_dan.py:299-404).
//
// An fp32 value is carried as TWO IEEE-half limbs, x = hi + lo with hi = half(x), lo = half(x - hi): 22 significand bits.  A product of
// two such numbers needs three half products, x.w ~ hi.hi + lo.hi + hi.lo (the dropped lo.lo term is 2^-22 relative), each EXACT in
// fp32 (11 x 11 bits) and accumulated in fp32 by v_mfma_f32_16x16x32_f16.  Laid out along the channel axis that IS an ordinary
// convolution over 3C input channels:
//     activations  X3 = [ hi | lo | hi ]   (NHWC, 3C halves per pixel, padded to a multiple of 8)
//     weights      W3 = [ hi | hi | lo ]   (HWIO [kh, kw, 3C, Cout], built once per variable on the host side: dan_amd/ops.py)
// so every 16-bit convolution kernel of the fp16 build (halo / pointwise / flat-M, with their fp32 accumulators and fp32 epilogue)
// computes it unchanged — three MFMAs per product instead of the sixteen v_mfma_f32_16x16x4_f32 of the fp32 path (csrc/f32_infer.hip).
// bf16 limbs would give 16 bits and need six products for three limbs; half limbs need the values inside half's range, which the
// detectors' activations are (the fp16 build runs the same graphs).  Subnormal limbs: the MFMA keeps them
// (tools/probe_f16_denorm.hip), so no scaling is applied.
//
// This file holds the conversions between fp32 maps and the 3C layout (the kernels are the same in both builds: IEEE half whatever the
// build's activation type is); the fused form — a convolution epilogue that writes the 3C layout directly — is conv_halo.hip's
// `split_out`.
#include "common.h"

namespace {

__device__ __forceinline__ void split_one(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);                     // x - hi is exact in fp32; one rounding to half
}

// One thread = 4 consecutive channels of one pixel (C % 4 == 0): one float4 load, three 8-byte stores.
__global__ __launch_bounds__(256) void split3_vec4_kernel(const float* __restrict__ x, _Float16* __restrict__ y, long M, int C, int C3, int relu) {
  const int c4 = C >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * c4) return;
  const long m = i / c4;
  const int c = (int)(i - m * c4) * 4;
  float4 v = *reinterpret_cast<const float4*>(x + m * C + c);
  if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  h4 hi, lo;
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    _Float16 a, b;
    split_one(vv[r], a, b);
    hi[r] = a; lo[r] = b;
  }
  _Float16* row = y + m * C3;
  *reinterpret_cast<h4*>(row + c) = hi;
  *reinterpret_cast<h4*>(row + C + c) = lo;
  *reinterpret_cast<h4*>(row + 2 * C + c) = hi;
}

// The 3-channel image (C3 = 16): one thread per pixel, the 16 halves [hi3 | lo3 | hi3 | 0 x 7] built in registers and stored as two
// 16-byte words (the generic kernel below wrote them as sixteen 2-byte stores: 0.2 ms per batch of 16 images at 640 x 640).
__global__ __launch_bounds__(256) void split3_c3_kernel(const float* __restrict__ x, _Float16* __restrict__ y, long M) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
  _Float16 hi[3], lo[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) split_one(x[m * 3 + c], hi[c], lo[c]);
  const _Float16 z = (_Float16)0.f;
  const h8 a = {hi[0], hi[1], hi[2], lo[0], lo[1], lo[2], hi[0], hi[1]}, b = {hi[2], z, z, z, z, z, z, z};
  h8* row = reinterpret_cast<h8*>(y + m * 16);
  row[0] = a;
  row[1] = b;
}

// Ragged channel counts: one thread per pixel; also zero-fills the padding columns [3C, C3).
__global__ __launch_bounds__(256) void split3_any_kernel(const float* __restrict__ x, _Float16* __restrict__ y, long M, int C, int C3, int relu) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  _Float16* row = y + m * C3;
  for (int c = 0; c < C; ++c) {
    float v = x[m * C + c];
    if (relu) v = fmaxf(v, 0.f);
    _Float16 hi, lo;
    split_one(v, hi, lo);
    row[c] = hi; row[C + c] = lo; row[2 * C + c] = hi;
  }
  for (int c = 3 * C; c < C3; ++c) row[c] = (_Float16)0.f;
}

__global__ __launch_bounds__(256) void unsplit3_kernel(const _Float16* __restrict__ x3, float* __restrict__ y, long M, int C, int C3) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * C) return;
  const long m = i / C;
  const int c = (int)(i - m * C);
  y[i] = (float)x3[m * C3 + c] + (float)x3[m * C3 + C + c];
}

// 2x2 / stride-2 'same' max-pool straight on the 3C layout (tf.layers.max_pooling2d of net/sfd_net.py:132; odd sizes: the window is
// clipped): the maximum of hi + lo over the window, re-split.  One thread = 4 consecutive channels of one OUTPUT pixel (C % 4 == 0).
__global__ __launch_bounds__(256) void maxpool2x2_split3_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int N, int H, int W, int C, int C3) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c4 = C >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * Ho * Wo * c4) return;
  const int c = (int)(i % c4) * 4;
  long p = i / c4;
  const int wo = (int)(p % Wo); p /= Wo;
  const int ho = (int)(p % Ho);
  const int n = (int)(p / Ho);
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int h = 2 * ho + dy, w = 2 * wo + dx;
      if (h < H && w < W) {
        const _Float16* row = x + ((long)(n * H + h) * W + w) * C3;
        const h4 a = *reinterpret_cast<const h4*>(row + c), b = *reinterpret_cast<const h4*>(row + C + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) best[r] = fmaxf(best[r], (float)a[r] + (float)b[r]);
      }
    }
  h4 hi, lo;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    _Float16 a, b;
    split_one(best[r], a, b);
    hi[r] = a; lo[r] = b;
  }
  _Float16* row = y + ((long)(n * Ho + ho) * Wo + wo) * C3;
  *reinterpret_cast<h4*>(row + c) = hi;
  *reinterpret_cast<h4*>(row + C + c) = lo;
  *reinterpret_cast<h4*>(row + 2 * C + c) = hi;
}

// l2_normalize (net/sfd_net.py:68-79) straight on the limb layout: one wave per pixel, the same lane-strided sum of squares, shuffle
// reduction and (x * inv) * gamma order as l2norm_f32_kernel (f32_infer.hip), so the result equals unsplit -> l2norm -> split to the limbs'
// precision — in one pass over 10 bytes per element instead of three passes over 26.
__global__ void l2norm_split3_kernel(const _Float16* __restrict__ x3, const float* __restrict__ gamma, _Float16* __restrict__ y3, long M, int C) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long m = wave; m < M; m += nw) {
    const _Float16* xp = x3 + m * 3 * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float v = (float)xp[c] + (float)xp[C + c];
      s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = rsqrtf(fmaxf(s, 1e-10f));
    _Float16* yp = y3 + m * 3 * C;
    for (int c = lane; c < C; c += 64) {
      const float v = (float)xp[c] + (float)xp[C + c];
      _Float16 hi, lo;
      split_one((v * inv) * gamma[c], hi, lo);
      yp[c] = hi; yp[C + c] = lo; yp[2 * C + c] = hi;
    }
  }
}

// The FIRST layer of the split-operand path (conv1_1: 3x3 / stride 1 / 'same', 3 input channels) computed directly in fp32 and stored in the
// next convolution's limb layout.  As limbs its operand has 9 channels padded to 16 (K = 144), which only the flat-M kernel's gather form takes:
// 1.15 ms per batch of 16 images at 640 x 640 plus 0.2 ms to split the image - for 0.7 GMAC per image.  With K = 27 the exact fp32 FMA chain
// costs less than the limb products' operand handling: a thread owns 4 horizontally adjacent pixels x 8 output channels (32 accumulators), the
// 3 x 6 x 3 input values it needs sit in registers, the 27 x Co weights in LDS (a lane's 8 channels = two ds_read_b128 per tap and channel);
// eight lanes write one pixel's 128-byte limb sections.  Bound by its 6 bytes per output element of stores.
template <int CO>
__global__ __launch_bounds__(256) void conv3x3_c3_f32_split3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                    _Float16* __restrict__ y3, int N, int H, int W, int relu) {
  constexpr int CG = CO / 8;                       // channel groups (lanes per pixel quad)
  __shared__ float ws[27 * CO];
  __shared__ float bs[CO];
  for (int i = threadIdx.x; i < 27 * CO; i += 256) ws[i] = w[i];          // HWIO: [(tap * 3 + c) * CO + co]
  for (int i = threadIdx.x; i < CO; i += 256) bs[i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int W4 = (W + 3) >> 2;
  const long quads = (long)N * H * W4;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  const long q = id / CG;
  const int cg = (int)(id - q * CG);
  if (q >= quads) return;
  const int x4 = (int)(q % W4) * 4;
  long r = q / W4;
  const int yy = (int)(r % H);
  const int n = (int)(r / H);
  float in[3][6][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int sy = yy - 1 + i;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int sx = x4 - 1 + j;
      const bool ok = (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
      const float* p = x + ((long)(n * H + (ok ? sy : 0)) * W + (ok ? sx : 0)) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) in[i][j][c] = ok ? p[c] : 0.f;
    }
  }
  float acc[4][8];
#pragma unroll
  for (int px = 0; px < 4; ++px)
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[px][o] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float4 wa = *reinterpret_cast<const float4*>(&ws[((i * 3 + j) * 3 + c) * CO + cg * 8]);
        const float4 wb = *reinterpret_cast<const float4*>(&ws[((i * 3 + j) * 3 + c) * CO + cg * 8 + 4]);
        const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
          for (int o = 0; o < 8; ++o) acc[px][o] = fmaf(in[i][px + j][c], wv[o], acc[px][o]);      // k order (tap, channel): conv_f32_kernel's
      }
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    if (x4 + px >= W) break;
    h8 hi, lo;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float v = acc[px][o] + bs[cg * 8 + o];
      if (relu) v = fmaxf(v, 0.f);
      _Float16 a, b;
      split_one(v, a, b);
      hi[o] = a; lo[o] = b;
    }
    _Float16* row = y3 + ((long)(n * H + yy) * W + x4 + px) * (3 * CO) + cg * 8;
    *reinterpret_cast<h8*>(row) = hi;
    *reinterpret_cast<h8*>(row + CO) = lo;
    *reinterpret_cast<h8*>(row + 2 * CO) = hi;
  }
}

}  // namespace

extern "C" int danhip_conv3x3_c3_f32_split3(const float* x, const float* w_hwio, const float* bias, uint16_t* y3, int32_t N, int32_t H, int32_t W,
                                            int32_t Cout, int relu, void* stream) {
  DH_REQUIRE(x && w_hwio && y3 && N > 0 && H > 0 && W > 0, DANHIP_EINVAL, "danhip_conv3x3_c3_f32_split3: bad arguments");
  DH_REQUIRE(Cout == 64, DANHIP_EINVAL, "danhip_conv3x3_c3_f32_split3: Cout = %d (the first layer of the VGG backbones: 64)", Cout);
  DH_REQUIRE((int64_t)N * H * W * 3 * Cout < (1ll << 40), DANHIP_EINVAL, "danhip_conv3x3_c3_f32_split3: too large");
  const long threads = (long)N * H * ((W + 3) / 4) * (Cout / 8);
  hipLaunchKernelGGL(conv3x3_c3_f32_split3_kernel<64>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w_hwio, bias,
                     reinterpret_cast<_Float16*>(y3), N, H, W, relu);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_l2norm_split3(const uint16_t* x3, const float* gamma, uint16_t* y3, int64_t M, int32_t C, void* stream) {
  DH_REQUIRE(x3 && gamma && y3 && M > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "danhip_l2norm_split3: bad arguments (C %% 8 == 0)");
  long blocks = (M * 64 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(l2norm_split3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(x3), gamma,
                     reinterpret_cast<_Float16*>(y3), (long)M, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_split3_f32(const float* x, uint16_t* y3, int64_t M, int32_t C, int32_t C3, int relu, void* stream) {
  DH_REQUIRE(x && y3 && M > 0 && C > 0, DANHIP_EINVAL, "danhip_split3_f32: bad arguments");
  DH_REQUIRE(C3 >= 3 * C && C3 % 8 == 0 && C3 < 3 * C + 8, DANHIP_EINVAL, "danhip_split3_f32: C3 = %d is not 3 * %d rounded up to 8", C3, C);
  hipStream_t s = (hipStream_t)stream;
  if (C == 3 && C3 == 16 && !relu) {
    hipLaunchKernelGGL(split3_c3_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, x, reinterpret_cast<_Float16*>(y3), (long)M);
  } else if (C % 4 == 0 && C3 == 3 * C) {
    const long n = M * (C / 4);
    hipLaunchKernelGGL(split3_vec4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, reinterpret_cast<_Float16*>(y3), (long)M, C, C3, relu);
  } else {
    hipLaunchKernelGGL(split3_any_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, x, reinterpret_cast<_Float16*>(y3), (long)M, C, C3, relu);
  }
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_unsplit3_f32(const uint16_t* x3, float* y, int64_t M, int32_t C, int32_t C3, void* stream) {
  DH_REQUIRE(x3 && y && M > 0 && C > 0 && C3 >= 3 * C, DANHIP_EINVAL, "danhip_unsplit3_f32: bad arguments");
  const long n = M * C;
  hipLaunchKernelGGL(unsplit3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(x3), y, (long)M, C, C3);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_maxpool2x2_split3(const uint16_t* x3, uint16_t* y3, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  DH_REQUIRE(x3 && y3 && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "danhip_maxpool2x2_split3: bad arguments (C %% 8 == 0)");
  const long n = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2x2_split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(x3), reinterpret_cast<_Float16*>(y3), N, H, W, C, 3 * C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}
