// HBM-bound layer kernels of the backbone / heads (gfx950): every access is a 16-byte (8 x bf16) vector per lane,
// reductions are wavefront (64-lane) shuffles.  Replaces the TF built-ins named at each entry point.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct bf8 { uint4 v; };
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack2bf(w[i], f[2 * i], f[2 * i + 1]);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 u;
  u.x = pack2bf(f[0], f[1]); u.y = pack2bf(f[2], f[3]); u.z = pack2bf(f[4], f[5]); u.w = pack2bf(f[6], f[7]);
  return u;
}

// ------------------------------------------------------------------ ReLU backward + bias gradient (one pass over dy)
// grid: (column groups of 8 channels) x (row chunks).  Each thread owns 8 channels and strides over rows.
__global__ void relu_bwd_bias_kernel(bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, float* __restrict__ db, long M, int C) {
  const int cg = C / 8;                                   // channel groups per row
  const int tpr = blockDim.x / cg > 0 ? blockDim.x / cg : 1;  // rows handled in parallel by one block
  const int g = threadIdx.x % cg, rsub = threadIdx.x / cg;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rsub < tpr) {
    for (long m = (long)blockIdx.x * tpr + rsub; m < M; m += (long)gridDim.x * tpr) {
      const long o = m * C + g * 8;
      uint4 d = *reinterpret_cast<const uint4*>(dy + o);
      float f[8];
      unpack8(d, f);
      if (y) {
        uint4 yy = *reinterpret_cast<const uint4*>(y + o);
        float yf[8];
        unpack8(yy, yf);
#pragma unroll
        for (int i = 0; i < 8; ++i) if (!(yf[i] > 0.f)) f[i] = 0.f;
        *reinterpret_cast<uint4*>(dy + o) = pack8(f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
  }
  if (!db) return;
  extern __shared__ float red[];                          // [blockDim.x][8]
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = s[i];
  __syncthreads();
  if (threadIdx.x < cg) {
    float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < tpr; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] += red[(r * cg + threadIdx.x) * 8 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(db + threadIdx.x * 8 + i, t[i]);
  }
}

// ------------------------------------------------------------------ 2x2/2 max pool, TF 'same'
// (arg != nullptr: also the 2-bit arg-max code per pooled element - first maximum in row-major window order, the rule of maxpool_bwd_kernel -
// [pooled pixel][C/4] bytes, channel c in bits 2(c%4).. of byte c/4: what maxpool_bwd_arg_kernel scatters through)
__global__ void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo,
                                   unsigned char* __restrict__ arg) {
  const int cg = C / 8;
  const long total = (long)N * Ho * Wo * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float m[8];
    unsigned codes = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int h = ho * 2 + dh, w = wo * 2 + dw;
        if (h < H && w < W) {
          float f[8];
          unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * H + h) * W + w) * C + g * 8), f);
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (f[i] > m[i]) { m[i] = f[i]; codes = (codes & ~(3u << (2 * i))) | ((unsigned)(dh * 2 + dw) << (2 * i)); }      // strict: the first maximum stays
        }
      }
    *reinterpret_cast<uint4*>(y + idx * 8) = pack8(m);
    if (arg) *reinterpret_cast<unsigned short*>(arg + (idx / cg) * (C / 4) + g * 2) = (unsigned short)codes;
  }
}

// backward: the gradient goes to the FIRST maximal element in window order (0,0),(0,1),(1,0),(1,1)
__global__ void maxpool_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N, int H, int W,
                                   int C, int Ho, int Wo, int accumulate) {
  const int cg = C / 8;
  const long total = (long)N * Ho * Wo * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float v[4][8];
    bool in[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int h = ho * 2 + (t >> 1), w = wo * 2 + (t & 1);
      in[t] = h < H && w < W;
      if (in[t]) unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * H + h) * W + w) * C + g * 8), v[t]);
      else
#pragma unroll
        for (int i = 0; i < 8; ++i) v[t][i] = -INFINITY;
    }
    float gy[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + idx * 8), gy);
    float o[4][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int best = 0;
      float bv = v[0][i];
#pragma unroll
      for (int t = 1; t < 4; ++t) if (v[t][i] > bv) { bv = v[t][i]; best = t; }
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t][i] = (t == best) ? gy[i] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int h = ho * 2 + (t >> 1), w = wo * 2 + (t & 1);
      if (in[t]) {
        uint4* dst = reinterpret_cast<uint4*>(dx + (((long)n * H + h) * W + w) * C + g * 8);
        if (accumulate) {
          float old[8];
          unpack8(*dst, old);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[t][i] += old[i];
        }
        *dst = pack8(o[t]);
      }
    }
  }
}

// ------------------------------------------------------------------ channel L2 normalisation with learned scale
// Every lane owns 8 consecutive channels (one 16-byte vector); a pixel is covered by LPP = C/8 lanes (8..64) of one wave
// (C = 1024: two 16-byte vectors per lane); per-pixel sums are xor-shuffle reductions over those lanes.
template <int LPP, int VPL>                        // lanes per pixel (power of two <= 64), vectors per lane
__device__ __forceinline__ float pixel_sum(float v) {
#pragma unroll
  for (int o = 1; o < LPP; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPP, int VPL>
__global__ void l2norm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, bf16_t* __restrict__ y, long M, int C) {
  constexpr int PPW = 64 / LPP;                    // pixels per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPP, l = lane % LPP;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  float gm[VPL][8];
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int i = 0; i < 8; ++i) gm[v][i] = gamma[(v * LPP + l) * 8 + i];
  for (long p0 = wave * PPW; p0 < M; p0 += nwaves * PPW) {
    const long pix = p0 + sub;
    const bool ok = pix < M;
    float f[VPL][8];
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      if (ok) unpack8(*reinterpret_cast<const uint4*>(x + pix * C + (v * LPP + l) * 8), f[v]);
      else
#pragma unroll
        for (int i = 0; i < 8; ++i) f[v][i] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += f[v][i] * f[v][i];
    }
    ss = pixel_sum<LPP, VPL>(ss);
    const float inv = rsqrtf(fmaxf(ss, 1e-10f));
    if (ok) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[v][i] = f[v][i] * inv * gm[v][i];
        *reinterpret_cast<uint4*>(y + pix * C + (v * LPP + l) * 8) = pack8(f[v]);
      }
    }
  }
}

// dx = gamma*inv*dy - x*inv^3 * sum_c(dy*gamma*x)   (second term dropped where sum x^2 <= 1e-10: clamp inactive grad)
// dgamma[c] += sum_pix dy*x*inv.  dx is ACCUMULATED into when `accumulate` (the tapped map also feeds the next conv block).
template <int LPP, int VPL>
__global__ void l2norm_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const bf16_t* __restrict__ dy,
                                  bf16_t* __restrict__ dx, float* __restrict__ dgamma, long M, int C, int accumulate, int relu_mask) {
  constexpr int PPW = 64 / LPP;
  extern __shared__ float sg[];                    // [C] block-level dgamma accumulator
  for (int c = threadIdx.x; c < C; c += blockDim.x) sg[c] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPP, l = lane % LPP;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  float gm[VPL][8], dg[VPL][8];
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int i = 0; i < 8; ++i) { gm[v][i] = gamma[(v * LPP + l) * 8 + i]; dg[v][i] = 0.f; }
  for (long p0 = wave * PPW; p0 < M; p0 += nwaves * PPW) {
    const long pix = p0 + sub;
    const bool ok = pix < M;
    float f[VPL][8], g[VPL][8];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const long o = pix * C + (v * LPP + l) * 8;
      if (ok) {
        unpack8(*reinterpret_cast<const uint4*>(x + o), f[v]);
        unpack8(*reinterpret_cast<const uint4*>(dy + o), g[v]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { f[v][i] = 0.f; g[v][i] = 0.f; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { ss += f[v][i] * f[v][i]; dot += g[v][i] * gm[v][i] * f[v][i]; }
    }
    ss = pixel_sum<LPP, VPL>(ss);
    dot = pixel_sum<LPP, VPL>(dot);
    const float inv = rsqrtf(fmaxf(ss, 1e-10f));
    const float k = (ss > 1e-10f) ? dot * inv * inv * inv : 0.f;
    if (ok) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const long o = pix * C + (v * LPP + l) * 8;
        float r[8], old[8];
        if (accumulate) unpack8(*reinterpret_cast<const uint4*>(dx + o), old);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float t = gm[v][i] * inv * g[v][i] - f[v][i] * k;
          if (relu_mask && !(f[v][i] > 0.f)) t = 0.f;          // x is a ReLU output: fold the producer's ReLU backward in
          if (accumulate) t += old[i];
          r[i] = t;
          dg[v][i] += g[v][i] * f[v][i] * inv;
        }
        *reinterpret_cast<uint4*>(dx + o) = pack8(r);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(sg + (v * LPP + l) * 8 + i, dg[v][i]);     // LDS atomics: PPW * waves adders per channel
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(dgamma + c, sg[c]);
}

// ------------------------------------------------------------------ input preprocessing
// uint8 RGB [N,H,W,3] -> bf16 [N,H,W,8] = (BGR - mean, 0,0,0,0,0)   (preprocessing/dan_preprocessing.py:55-57,755-758)
__global__ void preprocess_kernel(const unsigned char* __restrict__ img, bf16_t* __restrict__ out, long npix) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const float r = (float)img[p * 3] - 123.68f, g = (float)img[p * 3 + 1] - 116.78f, b = (float)img[p * 3 + 2] - 103.94f;
    float f[8] = {b, g, r, 0, 0, 0, 0, 0};
    *reinterpret_cast<uint4*>(out + p * 8) = pack8(f);
  }
}

__global__ void cast_pad_kernel(const float* __restrict__ src, const bf16_t* __restrict__ relu_y, bf16_t* __restrict__ dst, long rows, int c_src,
                                int c_dst) {
  const long total = rows * c_dst;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c_dst;
    const int c = (int)(i % c_dst);
    float v = c < c_src ? src[r * c_src + c] : 0.f;
    if (relu_y && c < c_src && !(bf2f(relu_y[r * c_src + c]) > 0.f)) v = 0.f;       // ReLU backward on the UNPADDED layout of y
    dst[i] = f2bf(v);
  }
}

inline int grid_for(long total, int block, int cap = 8192) {
  long b = (total + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int danhip_relu_bwd_bias_grad(uint16_t* dy, const uint16_t* y, float* db, int64_t M, int32_t C, void* stream) {
  DH_REQUIRE(dy && M > 0 && C > 0, DANHIP_EINVAL, "relu_bwd_bias_grad: bad arguments");
  DH_REQUIRE(C % 8 == 0 && C / 8 <= 256, DANHIP_EINVAL, "relu_bwd_bias_grad: C=%d must be a multiple of 8 and <= 2048", C);
  if (!y && !db) return DANHIP_OK;
  const int cg = C / 8;
  const int block = 256;
  const int tpr = block / cg > 0 ? block / cg : 1;
  long blocks = (M + tpr - 1) / tpr;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(relu_bwd_bias_kernel, dim3((unsigned)blocks), dim3(block), block * 8 * sizeof(float), (hipStream_t)stream, dy, y, db,
                     (long)M, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

namespace {
// ------------------------------------------------------------------ concat / residual-add backward: one input's slice of dY
// vector form: every row offset is a multiple of 8 channels -> one 16-byte lane access per 8 channels
__global__ void slice_deliver_vec_kernel(const bf16_t* __restrict__ dy, int ldy, int c0, int C, const bf16_t* __restrict__ mask, bf16_t* __restrict__ out,
                                         int accumulate, long M) {
  const int cg = C / 8;
  const long total = M * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / cg;
    const int g = (int)(idx - m * cg);
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + m * ldy + c0 + g * 8), f);
    if (mask) {
      float k[8];
      unpack8(*reinterpret_cast<const uint4*>(mask + idx * 8), k);
#pragma unroll
      for (int i = 0; i < 8; ++i) if (!(k[i] > 0.f)) f[i] = 0.f;
    }
    uint4* dst = reinterpret_cast<uint4*>(out + idx * 8);
    if (accumulate) {
      float old[8];
      unpack8(*dst, old);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += old[i];
    }
    *dst = pack8(f);
  }
}
// ragged form (C or c0 not a multiple of 8): one element per lane
__global__ void slice_deliver_elem_kernel(const bf16_t* __restrict__ dy, int ldy, int c0, int C, const bf16_t* __restrict__ mask, int ldm,
                                          bf16_t* __restrict__ out, int Cpad, int accumulate, long M) {
  const long total = M * Cpad;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / Cpad;
    const int c = (int)(idx - m * Cpad);
    float v = 0.f;
    if (c < C) {
      v = bf2f(dy[m * ldy + c0 + c]);
      if (mask && !(bf2f(mask[m * ldm + c]) > 0.f)) v = 0.f;
    }
    if (accumulate) v += bf2f(out[idx]);
    out[idx] = f2bf(v);
  }
}
}  // namespace

extern "C" int danhip_slice_deliver(const uint16_t* dy, int32_t ldy, int32_t c0, int32_t C, const uint16_t* mask, int32_t ldm, uint16_t* out,
                                    int32_t Cpad, int accumulate, int64_t M, void* stream) {
  DH_REQUIRE(dy && out && M > 0, DANHIP_EINVAL, "slice_deliver: bad arguments");
  DH_REQUIRE(C > 0 && c0 >= 0 && c0 + C <= ldy && Cpad >= C && Cpad % 8 == 0 && Cpad - C < 8, DANHIP_EINVAL,
             "slice_deliver: slice [%d, %d) of %d channels into rows of %d", c0, c0 + C, ldy, Cpad);
  DH_REQUIRE(!mask || ldm >= C, DANHIP_EINVAL, "slice_deliver: mask rows of %d < %d channels", ldm, C);
  const bool vec = C % 8 == 0 && c0 % 8 == 0 && ldy % 8 == 0 && (!mask || ldm == C);
  const long total = vec ? M * (C / 8) : M * Cpad;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (vec)
    hipLaunchKernelGGL(slice_deliver_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, ldy, c0, C, mask, out, accumulate, (long)M);
  else
    hipLaunchKernelGGL(slice_deliver_elem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, ldy, c0, C, mask, ldm, out, Cpad,
                       accumulate, (long)M);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

namespace {
// ------------------------------------------------------------------ ReLU mask as bits: bits[m][j] bit i = x[m][8j + i] > 0
// thread = 32 channels (four 16-byte loads -> one 32-bit store); C % 32 != 0: the tail groups store single bytes
__global__ void relu_bits_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ bits, long M, int C) {
  const int g32 = (C + 31) / 32, cb = C / 8;
  const long total = M * g32;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / g32;
    const int j = (int)(idx - m * g32);
    unsigned word = 0;
    const int nb = min(4, cb - j * 4);                 // bytes (8-channel groups) this thread owns
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nb) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + m * C + (j * 4 + b) * 8), f);
        unsigned byte = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) byte |= (f[i] > 0.f ? 1u : 0u) << i;
        word |= byte << (8 * b);
      }
    }
    unsigned char* dst = bits + m * cb + j * 4;
    if (nb == 4 && (cb & 3) == 0) *reinterpret_cast<unsigned*>(dst) = word;
    else
      for (int b = 0; b < nb; ++b) dst[b] = (unsigned char)(word >> (8 * b));
  }
}
}  // namespace

extern "C" int danhip_relu_bits(const uint16_t* x, uint8_t* bits, int64_t M, int32_t C, void* stream) {
  DH_REQUIRE(x && bits && M > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "relu_bits: bad arguments (C %% 8 == 0)");
  const long total = (long)M * ((C + 31) / 32);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(relu_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bits, (long)M, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

namespace {
__global__ void zero_fill_kernel(uint4* __restrict__ p16, long n16, unsigned* __restrict__ tail, int ntail) {
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p16[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}
}  // namespace

int danhip_zero_async(void* ptr, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return DANHIP_OK;
  DH_REQUIRE(ptr && ((uintptr_t)ptr & 15) == 0 && (bytes & 3) == 0, DANHIP_EINVAL, "zero_async: pointer must be 16-byte aligned, size a multiple of 4");
  const long n16 = (long)(bytes / 16);
  const int ntail = (int)((bytes % 16) / 4);
  long blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint4*)ptr, n16, (unsigned*)((char*)ptr + n16 * 16), ntail);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_maxpool2x2_fwd_arg(const uint16_t* x, uint16_t* y, uint8_t* arg, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  DH_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "maxpool2x2_fwd: bad arguments (C%%8)");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo, arg);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}
extern "C" int danhip_maxpool2x2_fwd(const uint16_t* x, uint16_t* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  return danhip_maxpool2x2_fwd_arg(x, y, nullptr, N, H, W, C, stream);
}

namespace {
// Max-pool backward THROUGH the arg-max codes (round 4): dx[2ho + dh][2wo + dw][c] (+)= dy[ho][wo][c] where the code of (ho, wo, c) is
// 2 dh + dw, 0 elsewhere in the window.  Reads dy (a quarter of the map) and 2 bits per pooled element instead of the full-resolution
// activation maxpool_bwd_kernel re-reads to find the maximum: 1.28 instead of 2.25 map-sized passes.
__global__ void maxpool_bwd_arg_kernel(const unsigned char* __restrict__ arg, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N, int H,
                                       int W, int C, int Ho, int Wo, int accumulate) {
  const int cg = C / 8;
  const long total = (long)N * Ho * Wo * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const long ppix = p;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const unsigned codes = *reinterpret_cast<const unsigned short*>(arg + ppix * (C / 4) + g * 2);
    float gy[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + idx * 8), gy);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int h = ho * 2 + (t >> 1), w = wo * 2 + (t & 1);
      if (h < H && w < W) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (((codes >> (2 * i)) & 3u) == (unsigned)t) ? gy[i] : 0.f;
        uint4* dst = reinterpret_cast<uint4*>(dx + (((long)n * H + h) * W + w) * C + g * 8);
        if (accumulate) {
          float old[8];
          unpack8(*dst, old);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += old[i];
        }
        *dst = pack8(o);
      }
    }
  }
}
}  // namespace

extern "C" int danhip_maxpool2x2_bwd_arg(const uint8_t* arg, const uint16_t* dy, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                         int accumulate, void* stream) {
  DH_REQUIRE(arg && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "maxpool2x2_bwd_arg: bad arguments (C%%8)");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_arg_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, arg, dy, dx, N, H, W, C, Ho, Wo, accumulate);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_maxpool2x2_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                     int accumulate, void* stream) {
  DH_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "maxpool2x2_bwd: bad arguments (C%%8)");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H, W, C, Ho, Wo, accumulate);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_l2norm_fwd(const uint16_t* x, const float* gamma, uint16_t* y, int64_t M, int32_t C, void* stream) {
  DH_REQUIRE(x && gamma && y && M > 0, DANHIP_EINVAL, "l2norm_fwd: bad arguments");
  DH_REQUIRE(C == 256 || C == 512 || C == 1024 || C == 128 || C == 64, DANHIP_EINVAL, "l2norm_fwd: C=%d unsupported", C);
  hipStream_t s = (hipStream_t)stream;
  const int lpp = C >= 512 ? 64 : C / 8, ppw = 64 / lpp;
  long blocks = (M + 4 * ppw - 1) / (4 * ppw);
  if (blocks > 4096) blocks = 4096;
  const dim3 g((unsigned)blocks), b(256);
  switch (C) {
    case 64: hipLaunchKernelGGL((l2norm_fwd_kernel<8, 1>), g, b, 0, s, x, gamma, y, (long)M, C); break;
    case 128: hipLaunchKernelGGL((l2norm_fwd_kernel<16, 1>), g, b, 0, s, x, gamma, y, (long)M, C); break;
    case 256: hipLaunchKernelGGL((l2norm_fwd_kernel<32, 1>), g, b, 0, s, x, gamma, y, (long)M, C); break;
    case 512: hipLaunchKernelGGL((l2norm_fwd_kernel<64, 1>), g, b, 0, s, x, gamma, y, (long)M, C); break;
    default: hipLaunchKernelGGL((l2norm_fwd_kernel<64, 2>), g, b, 0, s, x, gamma, y, (long)M, C); break;
  }
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_l2norm_bwd(const uint16_t* x, const float* gamma, const uint16_t* dy, uint16_t* dx, float* dgamma, int64_t M,
                                 int32_t C, int accumulate, int relu_mask, void* stream) {
  DH_REQUIRE(x && gamma && dy && dx && dgamma && M > 0, DANHIP_EINVAL, "l2norm_bwd: bad arguments");
  DH_REQUIRE(C == 256 || C == 512 || C == 1024 || C == 128 || C == 64, DANHIP_EINVAL, "l2norm_bwd: C=%d unsupported", C);
  hipStream_t s = (hipStream_t)stream;
  const int lpp = C >= 512 ? 64 : C / 8, ppw = 64 / lpp;
  // Every block ends with C atomics onto the SAME C addresses (dgamma), and contended float atomics run at ~0.09 TB/s
  // (MI355X_MICROARCH.md, global float atomics): 2048 blocks x 512 channels cost ~45 us whatever M was (measured: 46 / 68 us at batch 2).
  // 1024-thread blocks, two per CU (full occupancy), reduce over 16 waves in LDS first: 4x fewer contended atomics.
  long blocks = (M + 16 * ppw - 1) / (16 * ppw);
  if (blocks > 512) blocks = 512;
  const dim3 g((unsigned)blocks), b(1024);
  const size_t lds = (size_t)C * sizeof(float);
  switch (C) {
    case 64: hipLaunchKernelGGL((l2norm_bwd_kernel<8, 1>), g, b, lds, s, x, gamma, dy, dx, dgamma, (long)M, C, accumulate, relu_mask); break;
    case 128: hipLaunchKernelGGL((l2norm_bwd_kernel<16, 1>), g, b, lds, s, x, gamma, dy, dx, dgamma, (long)M, C, accumulate, relu_mask); break;
    case 256: hipLaunchKernelGGL((l2norm_bwd_kernel<32, 1>), g, b, lds, s, x, gamma, dy, dx, dgamma, (long)M, C, accumulate, relu_mask); break;
    case 512: hipLaunchKernelGGL((l2norm_bwd_kernel<64, 1>), g, b, lds, s, x, gamma, dy, dx, dgamma, (long)M, C, accumulate, relu_mask); break;
    default: hipLaunchKernelGGL((l2norm_bwd_kernel<64, 2>), g, b, lds, s, x, gamma, dy, dx, dgamma, (long)M, C, accumulate, relu_mask); break;
  }
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_preprocess_u8(const uint8_t* img_rgb, uint16_t* out, int64_t npix, void* stream) {
  DH_REQUIRE(img_rgb && out && npix > 0, DANHIP_EINVAL, "preprocess_u8: bad arguments");
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, img_rgb, out, (long)npix);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_cast_pad_f32_to_bf16(const float* src, const uint16_t* relu_y, uint16_t* dst, int64_t rows, int32_t c_src, int32_t c_dst,
                                           void* stream) {
  DH_REQUIRE(src && dst && rows > 0 && c_src > 0 && c_dst >= c_src, DANHIP_EINVAL, "cast_pad: bad arguments");
  hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for(rows * c_dst, 256)), dim3(256), 0, (hipStream_t)stream, src, relu_y, dst, (long)rows, c_src, c_dst);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}
