// HBM-bound layer kernels of the LFPN / DAN context modules (gfx950): 16-byte (8 x bf16) vectors per lane, fp32 math.
//   resize_bilinear (TF1 legacy, align_corners=False) fused with the lateral add   net/pb_net.py:209-217, net/danet.py:363-371
//   average_pooling2d((2,2), 1, 'same') with TF's valid-tap divisor                 net/danet.py:854
//   batch_normalization (training / inference / backward) for the conv_bn* surface  net/sfd_net.py:91-119
#include "common.h"

namespace {

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

// tf.image.resize_bilinear legacy source coordinate of destination index d: src = d * (in/out) in fp32
struct Lerp { int lo, hi; float w; };
__device__ __forceinline__ Lerp tf_legacy(int d, float scale, int in) {
  const float src = (float)d * scale;
  Lerp l;
  l.lo = (int)floorf(src);
  l.hi = min(l.lo + 1, in - 1);
  l.w = src - (float)l.lo;
  return l;
}

// out[n,ho,wo,:] = (lat ? lat[n,ho,wo,:] : 0) + bilinear(up)[n,ho,wo,:]
__global__ void resize_add_fwd_kernel(const bf16_t* __restrict__ up, const bf16_t* __restrict__ lat, bf16_t* __restrict__ out, int N, int Hi, int Wi,
                                      int Ho, int Wo, int C, float sh, float sw) {
  const int cg = C / 8;
  const long total = (long)N * Ho * Wo * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const Lerp ly = tf_legacy(ho, sh, Hi), lx = tf_legacy(wo, sw, Wi);
    float tl[8], tr[8], bl[8], br[8], r[8];
    const bf16_t* base = up + (long)n * Hi * Wi * C + g * 8;
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ly.lo * Wi + lx.lo) * C), tl);
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ly.lo * Wi + lx.hi) * C), tr);
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ly.hi * Wi + lx.lo) * C), bl);
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ly.hi * Wi + lx.hi) * C), br);
    if (lat) unpack8(*reinterpret_cast<const uint4*>(lat + idx * 8), r);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = tl[i] + (tr[i] - tl[i]) * lx.w;
      const float b = bl[i] + (br[i] - bl[i]) * lx.w;
      const float v = t + (b - t) * ly.w;
      r[i] = lat ? r[i] + v : v;
    }
    *reinterpret_cast<uint4*>(out + idx * 8) = pack8(r);
  }
}

// d_up[n,hi,wi,:] = sum over destination pixels that sample (hi,wi) of weight * dout  (deterministic gather form).
// Candidates: destinations d with floor(d*scale) in {i-1, i}; for an up-sampling (scale <= 1) that is at most
// ceil(2/scale)+1 per axis.
__global__ void resize_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ dup, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                  float sh, float sw, int accumulate) {
  const int cg = C / 8;
  const long total = (long)N * Hi * Wi * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int wi = (int)(p % Wi); p /= Wi;
    const int hi = (int)(p % Hi);
    const int n = (int)(p / Hi);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // destination range whose lo can be hi-1 or hi (conservative bounds, exact test inside)
    int h0 = (int)floorf((float)(hi - 1) / sh) - 1, h1 = (int)ceilf((float)(hi + 1) / sh) + 1;
    int w0 = (int)floorf((float)(wi - 1) / sw) - 1, w1 = (int)ceilf((float)(wi + 1) / sw) + 1;
    h0 = max(h0, 0); w0 = max(w0, 0); h1 = min(h1, Ho - 1); w1 = min(w1, Wo - 1);
    for (int ho = h0; ho <= h1; ++ho) {
      const Lerp ly = tf_legacy(ho, sh, Hi);
      float wy = 0.f;
      if (ly.lo == hi) wy += 1.f - ly.w;
      if (ly.hi == hi) wy += ly.w;
      if (wy == 0.f) continue;
      for (int wo = w0; wo <= w1; ++wo) {
        const Lerp lx = tf_legacy(wo, sw, Wi);
        float wx = 0.f;
        if (lx.lo == wi) wx += 1.f - lx.w;
        if (lx.hi == wi) wx += lx.w;
        if (wx == 0.f) continue;
        float gy[8];
        unpack8(*reinterpret_cast<const uint4*>(dout + (((long)n * Ho + ho) * Wo + wo) * C + g * 8), gy);
        const float wgt = wy * wx;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += wgt * gy[i];
      }
    }
    uint4* dst = reinterpret_cast<uint4*>(dup + idx * 8);
    if (accumulate) {
      float old[8];
      unpack8(*dst, old);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += old[i];
    }
    *dst = pack8(acc);
  }
}

// average_pooling2d((2,2), 1, 'same'): window (h..h+1, w..w+1) clipped to the map; divisor = number of valid taps
// (ldx / ldy: pixel pitches in elements - channel-slice views; relu: y = max(avg, 0), the DAN context block's branch 2 with its 1x1
// convolution moved in front of the pool)
__global__ void avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C, int ldx, int ldy, int relu) {
  const int cg = C / 8;
  const long total = (long)N * H * W * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        if (h + dh < H && w + dw < W) {
          float f[8];
          unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * H + h + dh) * W + w + dw) * ldx + g * 8), f);
#pragma unroll
          for (int i = 0; i < 8; ++i) s[i] += f[i];
          ++cnt;
        }
      }
    const float d = (float)cnt;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = s[i] / d;
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = fmaxf(s[i], 0.f);
    }
    *reinterpret_cast<uint4*>(y + (idx / cg) * ldy + g * 8) = pack8(s);
  }
}

__global__ void avgpool_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ xmask, bf16_t* __restrict__ dx, int N, int H, int W,
                                   int C, int accumulate, int ldy, int ldx) {
  const int cg = C / 8;
  const long total = (long)N * H * W * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int ho = h - dh, wo = w - dw;              // output whose window contains (h, w)
        if (ho >= 0 && wo >= 0) {
          const int cnt = ((ho + 1 < H) ? 2 : 1) * ((wo + 1 < W) ? 2 : 1);
          float f[8];
          unpack8(*reinterpret_cast<const uint4*>(dy + (((long)n * H + ho) * W + wo) * ldy + g * 8), f);
          const float d = (float)cnt;
#pragma unroll
          for (int i = 0; i < 8; ++i) s[i] += f[i] / d;
        }
      }
    uint4* dst = reinterpret_cast<uint4*>(dx + (idx / cg) * ldx + g * 8);
    if (xmask) {                                          // x is a ReLU output: fold its backward in (x > 0)
      float m[8];
      unpack8(*reinterpret_cast<const uint4*>(xmask + (idx / cg) * ldx + g * 8), m);
#pragma unroll
      for (int i = 0; i < 8; ++i) if (!(m[i] > 0.f)) s[i] = 0.f;
    }
    if (accumulate) {
      float old[8];
      unpack8(*dst, old);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += old[i];
    }
    *dst = pack8(s);
  }
}

// ------------------------------------------------------------------ batch normalisation over (N,H,W) per channel
// stats kernel: per-channel sum(a) and sum(a*b) over rows (b == nullptr: a*a); block = 64 channel-groups-of-8? No:
// thread owns 8 channels, strides rows; block-level reduction through LDS; one fp32 atomic per channel per block.
__global__ void bn_reduce_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ mean,
                                 const float* __restrict__ rstd, float* __restrict__ s1, float* __restrict__ s2, long M, int C) {
  // s1[c] += sum a ; s2[c] += sum a*a            (b == nullptr: forward statistics)
  // s1[c] += sum a ; s2[c] += sum a*(b-mean)*rstd (backward: a = dy, b = x)
  const int cg = C / 8;
  const int g = threadIdx.x % cg, rsub = threadIdx.x / cg, tpr = blockDim.x / cg;
  float t1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float mu[8], rs[8];
  if (b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { mu[i] = mean[g * 8 + i]; rs[i] = rstd[g * 8 + i]; }
  }
  if (rsub < tpr) {
    for (long r = (long)blockIdx.x * tpr + rsub; r < M; r += (long)gridDim.x * tpr) {
      float fa[8], fb[8];
      unpack8(*reinterpret_cast<const uint4*>(a + r * C + g * 8), fa);
      if (b) unpack8(*reinterpret_cast<const uint4*>(b + r * C + g * 8), fb);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        t1[i] += fa[i];
        t2[i] += b ? fa[i] * ((fb[i] - mu[i]) * rs[i]) : fa[i] * fa[i];
      }
    }
  }
  extern __shared__ float red[];
  float* r1 = red;
  float* r2 = red + blockDim.x * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r1[threadIdx.x * 8 + i] = t1[i]; r2[threadIdx.x * 8 + i] = t2[i]; }
  __syncthreads();
  if (threadIdx.x < cg) {
    for (int k = 1; k < tpr; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) { t1[i] += r1[(k * cg + threadIdx.x) * 8 + i]; t2[i] += r2[(k * cg + threadIdx.x) * 8 + i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { atomicAdd(s1 + threadIdx.x * 8 + i, t1[i]); atomicAdd(s2 + threadIdx.x * 8 + i, t2[i]); }
  }
}

// mean/var from the sums; optionally updates the moving averages: moving = moving*m + batch*(1-m).  tf.layers.batch_normalization on a
// 4-D input takes TF1's FUSED path, which normalises with the biased batch variance but feeds the moving average the Bessel-corrected
// one (var * M / (M - 1), nn_impl.fused_batch_norm's batch_var output)
__global__ void bn_finalize_kernel(const float* __restrict__ s1, const float* __restrict__ s2, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ var_out, float* __restrict__ mov_mean, float* __restrict__ mov_var, float inv_m, float eps,
                                   float momentum, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = s1[c] * inv_m;
  float var = s2[c] * inv_m - mu * mu;
  var = fmaxf(var, 0.f);
  mean[c] = mu;
  rstd[c] = rsqrtf(var + eps);
  if (var_out) var_out[c] = var;
  if (mov_mean) mov_mean[c] = mov_mean[c] * momentum + mu * (1.f - momentum);
  if (mov_var) {
    const float m = 1.f / inv_m;
    const float unbiased = m > 1.f ? var * (m / (m - 1.f)) : var;
    mov_var[c] = mov_var[c] * momentum + unbiased * (1.f - momentum);
  }
}

// y = (x - mean) * rstd * gamma + beta  (relu optional)
__global__ void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y, long M, int C, int relu) {
  const int cg = C / 8;
  const long total = M * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(x + idx * 8), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = g * 8 + i;
      float v = (f[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
      f[i] = relu ? fmaxf(v, 0.f) : v;
    }
    *reinterpret_cast<uint4*>(y + idx * 8) = pack8(f);
  }
}

// dx = gamma*rstd * (dy - mean(dy) - xhat*mean(dy*xhat));  sdy = sum dy, sdyx = sum dy*xhat (from bn_reduce_kernel)
__global__ void bn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ sdy,
                                    const float* __restrict__ sdyx, bf16_t* __restrict__ dx, long M, int C, float inv_m) {
  const int cg = C / 8;
  const long total = M * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    float fx[8], fd[8];
    unpack8(*reinterpret_cast<const uint4*>(x + idx * 8), fx);
    unpack8(*reinterpret_cast<const uint4*>(dy + idx * 8), fd);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = g * 8 + i;
      const float xh = (fx[i] - mean[c]) * rstd[c];
      fd[i] = gamma[c] * rstd[c] * (fd[i] - sdy[c] * inv_m - xh * sdyx[c] * inv_m);
    }
    *reinterpret_cast<uint4*>(dx + idx * 8) = pack8(fd);
  }
}

inline int grid_for(long total, int block = 256, int cap = 8192) {
  long b = (total + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int danhip_resize_bilinear_add_fwd(const uint16_t* up, const uint16_t* lateral, uint16_t* out, int32_t N, int32_t Hi, int32_t Wi,
                                              int32_t Ho, int32_t Wo, int32_t C, void* stream) {
  DH_REQUIRE(up && out, DANHIP_EINVAL, "resize_bilinear_add_fwd: null pointer");
  DH_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "resize_bilinear_add_fwd: bad dims (C %% 8 == 0)");
  const long total = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(resize_add_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, up, lateral, out, N, Hi, Wi, Ho, Wo, C,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_resize_bilinear_add_bwd(const uint16_t* dout, uint16_t* dup, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                              int32_t C, int accumulate, void* stream) {
  DH_REQUIRE(dout && dup, DANHIP_EINVAL, "resize_bilinear_add_bwd: null pointer");
  DH_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "resize_bilinear_add_bwd: bad dims (C %% 8 == 0)");
  const long total = (long)N * Hi * Wi * (C / 8);
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, dup, N, Hi, Wi, Ho, Wo, C,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo, accumulate);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

namespace {
// out = a + b (16-bit, 8 elements per thread): the context block's residual sum relu(conv(hyper)) + features (net/danet.py:913-918)
__global__ void add16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float fa[8], fb[8];
    unpack8(reinterpret_cast<const uint4*>(a)[i], fa);
    unpack8(reinterpret_cast<const uint4*>(b)[i], fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[e] += fb[e];
    reinterpret_cast<uint4*>(out)[i] = pack8(fa);
  }
}
// backward of y = r + x with r = relu(.) in ONE pass over dy:  dr = dy * (r > 0)   and   dx (+)= dy * (xmask > 0 | no mask)
__global__ void residual_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ r, const bf16_t* __restrict__ xmask, bf16_t* __restrict__ dr,
                                    bf16_t* __restrict__ dx, int accumulate, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float g[8], m[8], t[8];
    unpack8(reinterpret_cast<const uint4*>(dy)[i], g);
    unpack8(reinterpret_cast<const uint4*>(r)[i], m);
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = m[e] > 0.f ? g[e] : 0.f;
    reinterpret_cast<uint4*>(dr)[i] = pack8(t);
    if (dx) {
      if (xmask) {
        unpack8(reinterpret_cast<const uint4*>(xmask)[i], m);
#pragma unroll
        for (int e = 0; e < 8; ++e) if (!(m[e] > 0.f)) g[e] = 0.f;
      }
      if (accumulate) {
        unpack8(reinterpret_cast<const uint4*>(dx)[i], m);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += m[e];
      }
      reinterpret_cast<uint4*>(dx)[i] = pack8(g);
    }
  }
}
}  // namespace

extern "C" int danhip_add16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream) {
  DH_REQUIRE(a && b && out && n > 0 && n % 8 == 0, DANHIP_EINVAL, "add16: null pointer or n not a multiple of 8");
  hipLaunchKernelGGL(add16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)(n / 8));
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_residual_bwd(const uint16_t* dy, const uint16_t* r, const uint16_t* x_mask, uint16_t* dr, uint16_t* dx, int accumulate, int64_t n,
                                   void* stream) {
  DH_REQUIRE(dy && r && dr && n > 0 && n % 8 == 0, DANHIP_EINVAL, "residual_bwd: null pointer or n not a multiple of 8");
  hipLaunchKernelGGL(residual_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, dy, r, x_mask, dr, dx, accumulate, (long)(n / 8));
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_avgpool2x2s1_same_fwd(const uint16_t* x, uint16_t* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  DH_REQUIRE(x && y, DANHIP_EINVAL, "avgpool2x2s1_same_fwd: null pointer");
  DH_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "avgpool2x2s1_same_fwd: bad dims (C %% 8 == 0)");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, C, C, 0);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

/* The same pool on channel-slice views (x_pitch / y_pitch elements between pixels), optionally followed by ReLU; and its backward
 * (dy at y_pitch -> dx at x_pitch, overwritten; the ReLU backward is the caller's: dy arrives masked).  net/danet.py:854-861 with the
 * branch's 1x1 convolution commuted in front of the (linear) pool. */
extern "C" int danhip_avgpool2x2s1_same_fwd_strided(const uint16_t* x, int32_t x_pitch, uint16_t* y, int32_t y_pitch, int32_t N, int32_t H, int32_t W,
                                                    int32_t C, int relu, void* stream) {
  DH_REQUIRE(x && y, DANHIP_EINVAL, "avgpool2x2s1_same_fwd_strided: null pointer");
  DH_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && x_pitch >= C && y_pitch >= C && ((x_pitch | y_pitch) & 7) == 0, DANHIP_EINVAL,
             "avgpool2x2s1_same_fwd_strided: bad dims (C and the pitches multiples of 8, pitch >= C)");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, x_pitch, y_pitch,
                     relu ? 1 : 0);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_avgpool2x2s1_same_bwd_strided(const uint16_t* dy, int32_t y_pitch, uint16_t* dx, int32_t x_pitch, int32_t N, int32_t H, int32_t W,
                                                    int32_t C, void* stream) {
  DH_REQUIRE(dy && dx, DANHIP_EINVAL, "avgpool2x2s1_same_bwd_strided: null pointer");
  DH_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && x_pitch >= C && y_pitch >= C && ((x_pitch | y_pitch) & 7) == 0, DANHIP_EINVAL,
             "avgpool2x2s1_same_bwd_strided: bad dims (C and the pitches multiples of 8, pitch >= C)");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, dy, (const bf16_t*)nullptr, dx, N, H, W, C, 0,
                     y_pitch, x_pitch);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_avgpool2x2s1_same_bwd(const uint16_t* dy, const uint16_t* x_mask, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                            int accumulate, void* stream) {
  DH_REQUIRE(dy && dx, DANHIP_EINVAL, "avgpool2x2s1_same_bwd: null pointer");
  DH_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "avgpool2x2s1_same_bwd: bad dims (C %% 8 == 0)");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, dy, x_mask, dx, N, H, W, C, accumulate, C, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

namespace {
int bn_check(const void* x, int64_t M, int32_t C, const char* what) {
  DH_REQUIRE(x != nullptr, DANHIP_EINVAL, "%s: null pointer", what);
  DH_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, DANHIP_EINVAL, "%s: bad dims (C %% 8 == 0, C <= 2048)", what);
  return DANHIP_OK;
}
}  // namespace

/* workspace: 2*C floats (sums), zeroed inside */
extern "C" int danhip_batchnorm_fwd_train(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* save_mean,
                                          float* save_rstd, float* moving_mean, float* moving_var, int64_t M, int32_t C, float eps,
                                          float momentum, int relu, float* workspace, void* stream) {
  int rc = bn_check(x, M, C, "batchnorm_fwd_train");
  if (rc) return rc;
  DH_REQUIRE(gamma && beta && y && save_mean && save_rstd && workspace, DANHIP_EINVAL, "batchnorm_fwd_train: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, sizeof(float) * 2 * C, s) != hipSuccess) { danhip_set_error("batchnorm: memset failed"); return DANHIP_ELAUNCH; }
  const int cg = C / 8;
  const int block = cg <= 256 ? 256 / cg * cg : cg;        // multiple of cg (>= 1 row per block)
  const int tpr = block / cg;
  int grid = (int)((M + tpr - 1) / tpr);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(bn_reduce_kernel, dim3(grid), dim3(block), sizeof(float) * block * 16, s, x, (const bf16_t*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, workspace, workspace + C, (long)M, C);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, workspace, workspace + C, save_mean, save_rstd, (float*)nullptr,
                     moving_mean, moving_var, 1.0f / (float)M, eps, momentum, C);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * cg)), dim3(256), 0, s, x, save_mean, save_rstd, gamma, beta, y, (long)M, C, relu);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

/* rstd = rsqrt(moving_var + eps) must be supplied by the caller as `rstd` (precomputed once per evaluation) */
extern "C" int danhip_batchnorm_fwd_infer(const uint16_t* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                          uint16_t* y, int64_t M, int32_t C, int relu, void* stream) {
  int rc = bn_check(x, M, C, "batchnorm_fwd_infer");
  if (rc) return rc;
  DH_REQUIRE(gamma && beta && mean && rstd && y, DANHIP_EINVAL, "batchnorm_fwd_infer: null pointer");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, (long)M, C, relu);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

/* dgamma[c] = sum dy*xhat, dbeta[c] = sum dy (both OVERWRITTEN), dx as above.  workspace unused (sums land in dgamma/dbeta). */
extern "C" int danhip_batchnorm_bwd(const uint16_t* x, const uint16_t* dy, const float* gamma, const float* save_mean, const float* save_rstd,
                                    uint16_t* dx, float* dgamma, float* dbeta, int64_t M, int32_t C, void* stream) {
  int rc = bn_check(x, M, C, "batchnorm_bwd");
  if (rc) return rc;
  DH_REQUIRE(dy && gamma && save_mean && save_rstd && dx && dgamma && dbeta, DANHIP_EINVAL, "batchnorm_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dgamma, 0, sizeof(float) * C, s) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * C, s) != hipSuccess) {
    danhip_set_error("batchnorm_bwd: memset failed");
    return DANHIP_ELAUNCH;
  }
  const int cg = C / 8;
  const int block = cg <= 256 ? 256 / cg * cg : cg;
  const int tpr = block / cg;
  int grid = (int)((M + tpr - 1) / tpr);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(bn_reduce_kernel, dim3(grid), dim3(block), sizeof(float) * block * 16, s, dy, x, save_mean, save_rstd, dbeta, dgamma, (long)M, C);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * cg)), dim3(256), 0, s, x, dy, save_mean, save_rstd, gamma, dbeta, dgamma, dx, (long)M, C,
                     1.0f / (float)M);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

// ------------------------------------------------------------------------------------------------ 3x3 / stride-2 'same' max-pool
// tf.layers.max_pooling2d([3,3],[2,2],'same') — the ResNet stem's pool_1 (net/resnet_danet.py:129).  TF 'same': out = ceil(in/2),
// pad_before = max((out-1)*2 + 3 - in, 0) / 2; positions outside the image do not take part.  Backward (gather form, no atomics):
// an input pixel collects dy of every window whose FIRST maximum (window scan order) it is.
namespace {
__device__ __forceinline__ int pool3_pad(int in, int out) { int t = (out - 1) * 2 + 3 - in; return t > 0 ? t / 2 : 0; }

__global__ void maxpool3x3s2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
  const int cg = C / 8, pt = pool3_pad(H, Ho), pl = pool3_pad(W, Wo);
  const long total = (long)N * Ho * Wo * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
    for (int dh = 0; dh < 3; ++dh)
      for (int dw = 0; dw < 3; ++dw) {
        const int h = ho * 2 - pt + dh, w = wo * 2 - pl + dw;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
          float f[8];
          unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * H + h) * W + w) * C + g * 8), f);
#pragma unroll
          for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], f[i]);
        }
      }
    *reinterpret_cast<uint4*>(y + idx * 8) = pack8(m);
  }
}

__global__ void maxpool3x3s2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N, int H, int W,
                                        int C, int Ho, int Wo) {
  const int cg = C / 8, pt = pool3_pad(H, Ho), pl = pool3_pad(W, Wo);
  const long total = (long)N * H * W * cg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    long p = idx / cg;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // windows containing (h, w): ho with ho*2 - pt <= h <= ho*2 - pt + 2
    for (int ho = (h + pt - 2 + 1) / 2; ho <= (h + pt) / 2; ++ho) {
      if (ho < 0 || ho >= Ho) continue;
      for (int wo = (w + pl - 2 + 1) / 2; wo <= (w + pl) / 2; ++wo) {
        if (wo < 0 || wo >= Wo) continue;
        float best[8];
        int bidx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bidx[i] = -1; }
        int mine = -1;
        for (int dh = 0; dh < 3; ++dh)
          for (int dw = 0; dw < 3; ++dw) {
            const int hh = ho * 2 - pt + dh, ww = wo * 2 - pl + dw;
            if ((unsigned)hh >= (unsigned)H || (unsigned)ww >= (unsigned)W) continue;
            if (hh == h && ww == w) mine = dh * 3 + dw;
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * H + hh) * W + ww) * C + g * 8), f);
#pragma unroll
            for (int i = 0; i < 8; ++i) if (f[i] > best[i]) { best[i] = f[i]; bidx[i] = dh * 3 + dw; }
          }
        float gy[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + ((((long)n * Ho + ho) * Wo + wo) * C + g * 8)), gy);
#pragma unroll
        for (int i = 0; i < 8; ++i) if (bidx[i] == mine) acc[i] += gy[i];
      }
    }
    *reinterpret_cast<uint4*>(dx + idx * 8) = pack8(acc);
  }
}
}  // namespace

extern "C" int danhip_maxpool3x3s2_same_fwd(const uint16_t* x, uint16_t* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  DH_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "maxpool3x3s2_fwd: bad arguments (C%%8)");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_maxpool3x3s2_same_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                            void* stream) {
  DH_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DANHIP_EINVAL, "maxpool3x3s2_bwd: bad arguments (C%%8)");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * H * W * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H, W, C, Ho, Wo);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}
