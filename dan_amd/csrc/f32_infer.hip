// fp32 INFERENCE path (north_star: "eval_dan.py box outputs within 1e-4 of the reference on identical weights/inputs").
//
// The training path stores activations in 16 bits (bf16 / fp16): its logits differ from an fp32 reference by a few per cent of their
// scale after 20-60 layers, far from 1e-4 on the decoded boxes.  The evaluation graphs (eval_sfd.py:232-283, eval_pb.py, eval_dan.py:299-404)
// can therefore be run with fp32 storage and fp32 arithmetic end to end: NHWC fp32 activations, the TF variables used as they are (HWIO
// fp32 kernels, no packing), fp32-input MFMA (v_mfma_f32_16x16x4_f32: an exact k-ordered fmaf chain at the fp32 vector rate).  Same
// operator semantics as the 16-bit kernels (TF 'same' padding, first-max pooling, legacy bilinear resize, valid-tap average, the
// deformable sampling rules of cpp/Deform/deform_conv.cu:91-126,229-275).  Forward only.
//
// conv kernel: flat-M implicit GEMM, 64 pixels x 64 output channels per 256-thread workgroup, K walked in (tap, 16-channel) chunks
// through LDS (register-staged loads; the activation tile is gathered with the tap's shift and zero padding); each wave owns 16 pixels
// x 64 channels = four 16x16 accumulators.
#include "common.h"

namespace {

struct F32ConvArgs {
  const float* x;      // [N,H,W,C]
  const float* w;      // HWIO [kh,kw,C,Co]
  const float* bias;   // [Co] or null
  const float* resid;  // [M,Co] or null (added after the activation)
  float* y;            // [M,Co]
  int N, H, W, C, Ho, Wo, Co, kh, kw, stride, pad_t, pad_l, M, relu;
  FastDiv div_wo, div_howo;
};

__global__ __launch_bounds__(256) void conv_f32_kernel(const F32ConvArgs a) {
  constexpr int BM = 64, BN = 64, KC = 16;
  constexpr int AP = KC + 1;                           // padded rows: conflict-free column reads
  __shared__ float As[BM][AP];                         // [pixel][channel]
  __shared__ float Bs[KC][BN + 4];                     // [channel][co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  // staging roles: A: thread -> (pixel tid / 4, channels (tid % 4) * 4 .. +3); B: thread -> (k = tid / 16, co (tid % 16) * 4 .. +3)
  const int apx = tid >> 2, ac4 = (tid & 3) * 4;
  const int bk = tid >> 4, bco = (tid & 15) * 4;
  int an = 0, ah = -(1 << 20), aw = 0;                  // source pixel of tap (0,0) for this thread's staged row
  {
    const int m = m0 + apx;
    if (m < a.M) {
      const unsigned n = fdiv((unsigned)m, a.div_howo);
      const unsigned rem = (unsigned)m - n * (unsigned)(a.Ho * a.Wo);
      const unsigned ho = fdiv(rem, a.div_wo);
      an = (int)n; ah = (int)ho * a.stride - a.pad_t; aw = (int)(rem - ho * (unsigned)a.Wo) * a.stride - a.pad_l;
    }
  }
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int taps = a.kh * a.kw;
  const bool vec_a = (a.C & 3) == 0, vec_b = (a.Co & 3) == 0;
  for (int t = 0; t < taps; ++t) {
    const int ti = t / a.kw, tj = t - ti * a.kw;
    const int hy = ah + ti, wx = aw + tj;
    const bool pix_ok = (unsigned)hy < (unsigned)a.H && (unsigned)wx < (unsigned)a.W;
    const float* xp = a.x + ((size_t)(an * a.H + hy) * a.W + wx) * a.C;
    for (int c0 = 0; c0 < a.C; c0 += KC) {
      float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (pix_ok) {
        const int c = c0 + ac4;
        if (vec_a && c + 3 < a.C) {
          const float4 v = *reinterpret_cast<const float4*>(xp + c);
          av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < a.C) av[e] = xp[c + e];
        }
      }
      {
        const int c = c0 + bk, co = n0 + bco;
        if (c < a.C) {
          const float* wp = a.w + ((size_t)t * a.C + c) * a.Co + co;
          if (vec_b && co + 3 < a.Co) {
            const float4 v = *reinterpret_cast<const float4*>(wp);
            bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (co + e < a.Co) bv[e] = wp[e];
          }
        }
      }
      __syncthreads();                                  // previous chunk's fragments are in registers
#pragma unroll
      for (int e = 0; e < 4; ++e) As[apx][ac4 + e] = av[e];
#pragma unroll
      for (int e = 0; e < 4; ++e) Bs[bk][bco + e] = bv[e];
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        // v_mfma_f32_16x16x4_f32 operands: lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]
        const float af = As[wave * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bf = Bs[kk * 4 + (lane >> 4)][j * 16 + (lane & 15)];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // C/D layout: col = lane & 15 (co), row = (lane >> 4) * 4 + r (pixel)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = n0 + j * 16 + (lane & 15);
    if (co >= a.Co) continue;
    const float b = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + (lane >> 4) * 4 + r;
      if (m >= a.M) continue;
      float v = acc[j][r] + b;
      if (a.relu) v = fmaxf(v, 0.f);
      if (a.resid) v += a.resid[(size_t)m * a.Co + co];
      a.y[(size_t)m * a.Co + co] = v;
    }
  }
}

// ---- HBM-bound layers, one thread per output element (channel fastest: coalesced) ----------------------------------------------------
__global__ void maxpool2x2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m = -INFINITY;                                // tf.layers.max_pooling2d 'same': odd sizes pad with -inf (net/sfd_net.py:132)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int h = ho * 2 + dy, w = wo * 2 + dx;
        if (h < H && w < W) m = fmaxf(m, x[((long)(n * H + h) * W + w) * C + c]);
      }
    y[i] = m;
  }
}

// l2_normalize (net/sfd_net.py:68-79): one wave per pixel row
__global__ void l2norm_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma, float* __restrict__ y, long M, int C) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long m = wave; m < M; m += nw) {
    const float* xp = x + m * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xp[c] * xp[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = rsqrtf(fmaxf(s, 1e-10f));
    for (int c = lane; c < C; c += 64) y[m * C + c] = (xp[c] * inv) * gamma[c];
  }
}

// out = lateral + tf.image.resize_bilinear(up, size(out)) (TF1 legacy mapping: src = dst * in / out, hi = min(lo + 1, in - 1);
// net/pb_net.py:209-217) in the kernel's arithmetic order of the 16-bit version (top / bottom rows blended along x first, then y)
__global__ void resize_bilinear_add_f32_kernel(const float* __restrict__ up, const float* __restrict__ lat, float* __restrict__ out, int N, int Hi,
                                               int Wi, int Ho, int Wo, int C) {
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const long total = (long)N * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float fy = (float)y * sh, fx = (float)x * sw;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* b = up + (long)n * Hi * Wi * C + c;
    const float tl = b[((long)y0 * Wi + x0) * C], tr = b[((long)y0 * Wi + x1) * C];
    const float bl = b[((long)y1 * Wi + x0) * C], br = b[((long)y1 * Wi + x1) * C];
    const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
    const float v = top + (bot - top) * ly;
    out[i] = lat ? lat[i] + v : v;
  }
}

// tf.layers.average_pooling2d((2,2), 1, 'same') (net/danet.py:854): pad (0,1), divisor = number of valid taps
__global__ void avgpool2x2s1_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const long total = (long)N * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float s = 0.f;
    int cnt = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int hh = h + dy, ww = w + dx;
        if (hh < H && ww < W) { s += x[((long)(n * H + hh) * W + ww) * C + c]; ++cnt; }
      }
    y[i] = s / (float)cnt;
  }
}

// Deformable im2col (cpp/Deform/deform_conv.cu:229-275, :91-126) in fp32: S[m][t*C + c]; one thread per (m, t, c)
__global__ void deform_sample_f32_kernel(const float* __restrict__ x, const float* __restrict__ offs, float* __restrict__ S, int N, int H, int W, int C,
                                         int Ho, int Wo, int kh, int kw, int stride, int dil, int dg, int pad_t, int pad_l) {
  const int taps = kh * kw, cpg = C / dg, offc = dg * 2 * taps;
  const long total = (long)N * Ho * Wo * taps * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int t = (int)(r % taps);
    const long m = r / taps;
    const int wo = (int)(m % Wo);
    const int ho = (int)((m / Wo) % Ho);
    const int n = (int)(m / ((long)Wo * Ho));
    const int grp = c / cpg;
    const int i = t / kw, j = t % kw;
    const int h_in = ho * stride - pad_t, w_in = wo * stride - pad_l;
    const float* op = offs + m * offc + (grp * taps + t) * 2;
    const float off_h = op[0], off_w = op[1];
    const float h_im = (float)(h_in + i * dil) + off_h;             // deform_conv.cu:261-262
    const float w_im = (float)(w_in + j * dil) + off_w;
    float out = 0.f;
    if (h_im >= 0 && w_im >= 0 && h_im < H && w_im < W) {           // :263
      float mh = (float)(i * dil) + off_h, mw = (float)(j * dil) + off_w;
      const int cur_h = H - h_in, cur_w = W - w_in;
      int h_low = (int)floorf(mh), w_low = (int)floorf(mw), h_high, w_high;
      if (h_low >= cur_h - 1) { h_high = h_low = cur_h - 1; mh = (float)h_low; } else h_high = h_low + 1;
      if (w_low >= cur_w - 1) { w_high = w_low = cur_w - 1; mw = (float)w_low; } else w_high = w_low + 1;
      const float lh = mh - h_low, lw = mw - w_low, hh = 1 - lh, hw = 1 - lw;
      const float* base = x + ((long)n * H * W) * C + c;
      const float v1 = base[((long)(h_in + h_low) * W + (w_in + w_low)) * C], v2 = base[((long)(h_in + h_low) * W + (w_in + w_high)) * C];
      const float v3 = base[((long)(h_in + h_high) * W + (w_in + w_low)) * C], v4 = base[((long)(h_in + h_high) * W + (w_in + w_high)) * C];
      out = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
    }
    S[(m * taps + t) * C + c] = out;
  }
}

inline int f32_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}
inline int f32_same_pad_before(int in, int out, int k, int s) {
  int total = (out - 1) * s + k - in;
  if (total < 0) total = 0;
  return total / 2;
}

}  // namespace

extern "C" int danhip_conv2d_fwd_f32(const danhip_conv_desc* d, const float* x, const float* w_hwio, const float* bias, float* y, int relu,
                                     const float* residual, void* stream) {
  DH_REQUIRE(d && x && w_hwio && y, DANHIP_EINVAL, "conv2d_fwd_f32: null pointer");
  DH_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh >= 1 && d->kw >= 1 && d->stride >= 1, DANHIP_EINVAL,
             "conv2d_fwd_f32: non-positive dims");
  {
    const bool same = d->Ho == (d->H + d->stride - 1) / d->stride && d->Wo == (d->W + d->stride - 1) / d->stride;
    const bool valid = d->H >= d->kh && d->W >= d->kw && d->Ho == (d->H - d->kh) / d->stride + 1 && d->Wo == (d->W - d->kw) / d->stride + 1;
    DH_REQUIRE(same || valid, DANHIP_EINVAL, "conv2d_fwd_f32: Ho/Wo (%d,%d) is neither the 'same' nor the 'valid' output size", d->Ho, d->Wo);
  }
  DH_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (1ll << 31), DANHIP_EINVAL, "conv2d_fwd_f32: too many output pixels");
  F32ConvArgs a{};
  a.x = x; a.w = w_hwio; a.bias = bias; a.resid = residual; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Cout; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
  a.pad_t = f32_same_pad_before(d->H, d->Ho, d->kh, d->stride);
  a.pad_l = f32_same_pad_before(d->W, d->Wo, d->kw, d->stride);
  a.M = d->N * d->Ho * d->Wo;
  a.relu = relu;
  a.div_wo = make_fastdiv(a.Wo);
  a.div_howo = make_fastdiv(a.Ho * a.Wo);
  hipLaunchKernelGGL(conv_f32_kernel, dim3((a.M + 63) / 64, (a.Co + 63) / 64), dim3(256), 0, (hipStream_t)stream, a);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_maxpool2x2_fwd_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  DH_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, DANHIP_EINVAL, "maxpool2x2_fwd_f32: bad arguments");
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
  hipLaunchKernelGGL(maxpool2x2_f32_kernel, dim3(f32_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_l2norm_fwd_f32(const float* x, const float* gamma, float* y, int64_t M, int32_t C, void* stream) {
  DH_REQUIRE(x && gamma && y && M > 0 && C > 0, DANHIP_EINVAL, "l2norm_fwd_f32: bad arguments");
  hipLaunchKernelGGL(l2norm_f32_kernel, dim3(f32_blocks(M * 64)), dim3(256), 0, (hipStream_t)stream, x, gamma, y, (long)M, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_resize_bilinear_add_fwd_f32(const float* up, const float* lateral, float* out, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho,
                                                  int32_t Wo, int32_t C, void* stream) {
  DH_REQUIRE(up && out && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, DANHIP_EINVAL, "resize_bilinear_add_fwd_f32: bad arguments");
  hipLaunchKernelGGL(resize_bilinear_add_f32_kernel, dim3(f32_blocks((long)N * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, up, lateral, out, N, Hi,
                     Wi, Ho, Wo, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_avgpool2x2s1_same_fwd_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  DH_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, DANHIP_EINVAL, "avgpool2x2s1_same_fwd_f32: bad arguments");
  hipLaunchKernelGGL(avgpool2x2s1_f32_kernel, dim3(f32_blocks((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_deform_sample_fwd_f32(const float* x, const float* offsets, float* S, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh,
                                            int32_t kw, int32_t stride, int32_t dilation, int32_t deformable_group, void* stream) {
  DH_REQUIRE(x && offsets && S, DANHIP_EINVAL, "deform_sample_fwd_f32: null pointer");
  DH_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && dilation > 0 && deformable_group > 0 && C % deformable_group == 0,
             DANHIP_EINVAL, "deform_sample_fwd_f32: bad dims");
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  // pad from the UNDILATED kernel, as the reference (deform_conv.cc:473-479)
  const int pad_t = f32_same_pad_before(H, Ho, kh, stride), pad_l = f32_same_pad_before(W, Wo, kw, stride);
  hipLaunchKernelGGL(deform_sample_f32_kernel, dim3(f32_blocks((long)N * Ho * Wo * kh * kw * C)), dim3(256), 0, (hipStream_t)stream, x, offsets, S, N,
                     H, W, C, Ho, Wo, kh, kw, stride, dilation, deformable_group, pad_t, pad_l);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}
