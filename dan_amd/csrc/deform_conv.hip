// Deformable convolution sampling for gfx950 — the arithmetic of cpp/Deform/deform_conv.cu (deformable_im2col_bilinear
// :91-126, deformable_im2col_gpu_kernel :229-275, deformable_col2im_coord_gpu_kernel :335-389 + get_coordinate_weight
// :177-221, deformable_col2im_gpu_kernel :281-328 + get_gradient_weight :130-173) restated for NHWC bf16 tensors.
//
// DeformConvOp (cpp/Deform/deform_conv.cc:392-535) = per sample { deformable im2col ; GEMM with W[Cout, Cin*kh*kw] }.
// Here the im2col is the batched kernel below writing S[pixel][tap*C + c] (bf16, K-contiguous), and the GEMM is the MFMA
// convolution kernel run as a 1x1 convolution over S (forward, data gradient = dS, weight gradient) — see
// dan_amd/ops.py:deform_conv_op.  DeformConvBackpropOp (:635-771): dS = dY * W (1x1 data gradient), then
// deform_sample_bwd produces dOffset (gather over the group's channels) and dX (bilinear scatter, fp32 atomics as the
// reference does with CudaAtomicAdd :314-326).
//
// Offsets: NHWC [B,Ho,Wo, dg*2*kh*kw], channel (g*kh*kw + t)*2 + {0: dh, 1: dw}  (deform_conv.cu:251-259).
#include "common.h"

namespace {

struct DeformGeom {
  int N, H, W, C, Ho, Wo, kh, kw, stride, dil, dg, pad_t, pad_l;
};

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

// A 256-thread block owns SAMPLE_PXB consecutive output pixels of ONE output row: (image, row, first column) are block-uniform
// (scalar registers) and a thread's item q -> (pixel in the block, tap, 8-channel chunk) is 32-bit arithmetic — the earlier form (one
// flat 64-bit index per thread, five 64-bit divisions each) spent more instructions on its index than on the four 16-byte gathers.
// Items run chunk-fastest, then tap: a wave stores 1 KiB of S contiguously and its corner loads are whole 128-byte group rows.
constexpr int SAMPLE_PXB = 8;
__global__ __launch_bounds__(256) void deform_sample_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ offs, bf16_t* __restrict__ S,
                                                                DeformGeom g, int segs, int cch_shift) {
  const int taps = g.kh * g.kw, cch = g.C / 8, cpg8 = g.C / g.dg / 8;
  const int offc = g.dg * 2 * taps;
  const int seg = (int)(blockIdx.x % (unsigned)segs);
  const int rowid = (int)(blockIdx.x / (unsigned)segs);                 // n * Ho + ho
  const int ho = rowid % g.Ho, n = rowid / g.Ho;
  const int wo0 = seg * SAMPLE_PXB;
  const int npx = min(SAMPLE_PXB, g.Wo - wo0);
  const int items = npx * taps * cch;
  const int h_in = ho * g.stride - g.pad_t;
  const long m0 = (long)rowid * g.Wo + wo0;
  const bf16_t* xn = x + ((long)n * g.H * g.W) * g.C;
  const int cur_h = g.H - h_in;
  // Three items per trip: the three offset words are fetched first, then all twelve corner rows are in flight together — a trip
  // costs two dependent memory latencies instead of six (the kernel is latency bound, not bandwidth bound: 16 bytes per lane and load).
  constexpr int U = 3;
  for (int q0 = threadIdx.x; q0 < items; q0 += 256 * U) {
    int chunk[U], t[U], pxl[U], grp[U];
    bool live[U];
    unsigned oraw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int q = q0 + u * 256;
      live[u] = q < items;
      if (!live[u]) q = items - 1;
      int r;
      if (cch_shift >= 0) { chunk[u] = q & (cch - 1); r = q >> cch_shift; } else { chunk[u] = q % cch; r = q / cch; }
      pxl[u] = r / taps; t[u] = r - pxl[u] * taps;
      grp[u] = chunk[u] / cpg8;
      oraw[u] = *reinterpret_cast<const unsigned*>(offs + (m0 + pxl[u]) * offc + (grp[u] * taps + t[u]) * 2);
    }
    uint4 c4[U][4];
    float wq[U][4];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = t[u] / g.kw, j = t[u] - i * g.kw;
      const int w_in = (wo0 + pxl[u]) * g.stride - g.pad_l;
      const float off_h = bf2f((bf16_t)(oraw[u] & 0xffffu)), off_w = bf2f((bf16_t)(oraw[u] >> 16));
      const float h_im = (float)(h_in + i * g.dil) + off_h;             // deform_conv.cu:261-262
      const float w_im = (float)(w_in + j * g.dil) + off_w;
      in[u] = h_im >= 0 && w_im >= 0 && h_im < g.H && w_im < g.W;       // :263
      float mh = (float)(i * g.dil) + off_h, mw = (float)(j * g.dil) + off_w;     // :264-265 (relative to (h_in, w_in))
      if (!in[u]) { mh = (float)(-h_in); mw = (float)(-w_in); }         // (any valid address: the result is discarded)
      const int cur_w = g.W - w_in;                                     // :266-268
      int h_low = (int)floorf(mh), w_low = (int)floorf(mw), h_high, w_high;       // deformable_im2col_bilinear :94-112
      if (h_low >= cur_h - 1) { h_high = h_low = cur_h - 1; mh = (float)h_low; } else h_high = h_low + 1;
      if (w_low >= cur_w - 1) { w_high = w_low = cur_w - 1; mw = (float)w_low; } else w_high = w_low + 1;
      const float lh = mh - h_low, lw = mw - w_low, hh = 1 - lh, hw = 1 - lw;
      wq[u][0] = hh * hw; wq[u][1] = hh * lw; wq[u][2] = lh * hw; wq[u][3] = lh * lw;         // :118-125
      const bf16_t* base = xn + chunk[u] * 8;
      const int rl = (h_in + h_low) * g.W + w_in, rh = (h_in + h_high) * g.W + w_in;      // pixel indices inside image n
      c4[u][0] = *reinterpret_cast<const uint4*>(base + (long)(rl + w_low) * g.C);
      c4[u][1] = *reinterpret_cast<const uint4*>(base + (long)(rl + w_high) * g.C);
      c4[u][2] = *reinterpret_cast<const uint4*>(base + (long)(rh + w_low) * g.C);
      c4[u][3] = *reinterpret_cast<const uint4*>(base + (long)(rh + w_high) * g.C);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v1[8], v2[8], v3[8], v4[8], out[8];
      unpack8(c4[u][0], v1); unpack8(c4[u][1], v2); unpack8(c4[u][2], v3); unpack8(c4[u][3], v4);
#pragma unroll
      for (int e = 0; e < 8; ++e)       // (explicit fma chain: the fused kernel of deform_fused.hip must round the same way)
        out[e] = in[u] ? fmaf(wq[u][3], v4[e], fmaf(wq[u][2], v3[e], fmaf(wq[u][1], v2[e], wq[u][0] * v1[e]))) : 0.f;
      if (live[u]) *reinterpret_cast<uint4*>(S + ((m0 + pxl[u]) * taps + t[u]) * g.C + chunk[u] * 8) = pack8(out);
    }
  }
}

// backward: dS [M][taps*C] -> dOffset [M][dg*2*taps] (bf16, written), dX fp32 [N,H,W,C] (atomic +=, zero it first)
// thread = (m, t, 8-channel chunk); the chunks of one deformable group are consecutive lanes (cpg8 a power of two <= 32)
__global__ void deform_sample_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ offs, const bf16_t* __restrict__ dS,
                                         float* __restrict__ dx, bf16_t* __restrict__ doffs, DeformGeom g) {
  const int taps = g.kh * g.kw, cch = g.C / 8, cpg8 = g.C / g.dg / 8;
  const long total = (long)g.N * g.Ho * g.Wo * taps * cch;
  const int offc = g.dg * 2 * taps;
  const long total_pad = (total + blockDim.x - 1) / blockDim.x * blockDim.x;        // keep whole waves alive for the shuffles
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_pad; idx += (long)gridDim.x * blockDim.x) {
    const bool live = idx < total;
    const long id = live ? idx : total - 1;
    const int chunk = (int)(id % cch);
    long r = id / cch;
    const int t = (int)(r % taps);
    const long m = r / taps;
    const int wo = (int)(m % g.Wo);
    const int ho = (int)((m / g.Wo) % g.Ho);
    const int n = (int)(m / ((long)g.Wo * g.Ho));
    const int grp = chunk / cpg8;
    const int i = t / g.kw, j = t % g.kw;
    const int h_in = ho * g.stride - g.pad_t, w_in = wo * g.stride - g.pad_l;
    const bf16_t* op = offs + m * offc + (grp * taps + t) * 2;
    const float inv_h = (float)(h_in + i * g.dil) + bf2f(op[0]);      // absolute sample coordinate (:365-366, :304-305)
    const float inv_w = (float)(w_in + j * g.dil) + bf2f(op[1]);
    float cg[8];
    unpack8(*reinterpret_cast<const uint4*>(dS + (m * taps + t) * g.C + chunk * 8), cg);
    const bf16_t* base = x + ((long)n * g.H * g.W) * g.C + chunk * 8;
    const float Hf = (float)g.H, Wf = (float)g.W;
    // ---- dOffset: get_coordinate_weight (:177-221); out-of-range samples give 0 (:371-373)
    float s_h = 0.f, s_w = 0.f;
    if (!(inv_h < 0 || inv_w < 0 || inv_h >= Hf || inv_w >= Wf)) {
      float ih = inv_h, iw = inv_w;
      int h_low = (int)ih, w_low = (int)iw, h_high, w_high;           // (int) truncation as in the reference
      if (h_low >= g.H - 1) { h_high = h_low = g.H - 1; ih = (float)h_low; } else h_high = h_low + 1;
      if (w_low >= g.W - 1) { w_high = w_low = g.W - 1; iw = (float)w_low; } else w_high = w_low + 1;
      float vll[8], vlh[8], vhl[8], vhh[8];
      unpack8(*reinterpret_cast<const uint4*>(base + ((long)h_low * g.W + w_low) * g.C), vll);
      unpack8(*reinterpret_cast<const uint4*>(base + ((long)h_low * g.W + w_high) * g.C), vlh);
      unpack8(*reinterpret_cast<const uint4*>(base + ((long)h_high * g.W + w_low) * g.C), vhl);
      unpack8(*reinterpret_cast<const uint4*>(base + ((long)h_high * g.W + w_high) * g.C), vhh);
      const float a_w = (float)(w_low + 1) - iw, b_w = iw - (float)w_low;
      const float a_h = (float)(h_low + 1) - ih, b_h = ih - (float)h_low;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float wgt_h = -1.f * a_w * vll[e] + -1.f * b_w * vlh[e] + a_w * vhl[e] + b_w * vhh[e];
        const float wgt_w = -1.f * a_h * vll[e] + a_h * vlh[e] + -1.f * b_h * vhl[e] + b_h * vhh[e];
        s_h += wgt_h * cg[e];
        s_w += wgt_w * cg[e];
      }
    }
    if (!live) { s_h = 0.f; s_w = 0.f; }
    for (int o = 1; o < cpg8; o <<= 1) { s_h += __shfl_xor(s_h, o); s_w += __shfl_xor(s_w, o); }
    if (live && (chunk % cpg8) == 0) {
      bf16_t* dp = doffs + m * offc + (grp * taps + t) * 2;
      dp[0] = f2bf(s_h);
      dp[1] = f2bf(s_w);
    }
    // ---- dX: get_gradient_weight (:130-173) scattered to the <= 4 corners (:311-326)
    if (live && !(inv_h < 0 || inv_h > Hf || inv_w < 0 || inv_w > Wf)) {
      float ah = fmaxf(inv_h, 0.f), aw = fmaxf(inv_w, 0.f);
      int hl = (int)ah, wl = (int)aw, hh, wh;
      const bool ch = hl >= g.H - 1, cw = wl >= g.W - 1;
      if (ch) { hh = hl = g.H - 1; ah = (float)hl; } else hh = hl + 1;
      if (cw) { wh = wl = g.W - 1; aw = (float)wl; } else wh = wl + 1;
      const int rh[4] = {hl, hl, hh, hh}, rw[4] = {wl, wh, wl, wh};
      const float wg[4] = {((float)(hl + 1) - ah) * ((float)(wl + 1) - aw), ((float)(hl + 1) - ah) * (aw + 1.f - (float)wh),
                           (ah + 1.f - (float)hh) * ((float)(wl + 1) - aw), (ah + 1.f - (float)hh) * (aw + 1.f - (float)wh)};
      const bool dup[4] = {false, cw, ch, ch || cw};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = !dup[k] && fabsf(inv_h - (float)rh[k]) < 1.f && fabsf(inv_w - (float)rw[k]) < 1.f && rh[k] >= 0 && rh[k] < g.H &&
                        rw[k] >= 0 && rw[k] < g.W;
        if (ok) {
          float* dst = dx + (((long)n * g.H + rh[k]) * g.W + rw[k]) * g.C + chunk * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) atomicAdd(dst + e, wg[k] * cg[e]);
        }
      }
    }
  }
}


// ---- which backward form runs is decided ON THE DEVICE (no host synchronisation, capturable): the gather form below enumerates source
// taps within +-DEFORM_R of an input pixel and hands every corner outside that window to fp32 atomics one by one ("far" corners) — 3.2 ms
// at 160 x 160 x 256 with offsets of N(0, 0.5 px), but 43 ms at N(0, 2 px), where 85 % of the taps have a far corner; the all-atomics
// scatter form costs 12 ms whatever the offsets are.  deform_far_stat_kernel counts the (dh, dw) pairs outside [-2, 2) and [-1, 1) into
// stat[0..1] (zeroed with the scatter buffer); every backward kernel reads them and returns at once when it belongs to another form.
struct BwdGate {
  const unsigned* stat;   // stat[0] / stat[1] = number of offset pairs with a component outside [-2, 2) / [-1, 1)
  unsigned thresh2;       // scatter form when stat[0] > thresh2
  unsigned thresh1;       // gather window +-1 (81 candidates instead of 225) when stat[1] <= thresh1
  int force;              // option "deform_bwd_form": 0 by the statistics, 1 gather +-2, 2 scatter, 3 gather +-1
};
// form: 0 = gather with a +-1 window, 1 = gather with a +-2 window, 2 = scatter
__device__ __forceinline__ int gate_form(const BwdGate& q) {
  if (q.force) return q.force == 3 ? 0 : q.force;
  if (q.stat[0] > q.thresh2) return 2;
  return q.stat[1] <= q.thresh1 ? 0 : 1;
}
__device__ __forceinline__ bool gate_runs(const BwdGate& q, int form) { return gate_form(q) == form; }

// Fast path for C / deformable_group == 64: one WAVE per (output pixel, tap, deformable group), lane = channel.  Every
// atomic wave-instruction then adds 64 consecutive floats (256 contiguous bytes — the full-rate shape of the memory-side
// float atomics), and the offset gradient is a plain 64-lane reduction.
__global__ void deform_sample_bwd_c64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ offs, const bf16_t* __restrict__ dS,
                                             float* __restrict__ dx, bf16_t* __restrict__ doffs, DeformGeom g, BwdGate gate) {
  if (gate.stat && !gate_runs(gate, 2)) return;
  const int taps = g.kh * g.kw;
  const int offc = g.dg * 2 * taps;
  const int lane = threadIdx.x & 63;
  const long nwork = (long)g.N * g.Ho * g.Wo * taps * g.dg;              // (m, t, grp) triples, grp fastest
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const float Hf = (float)g.H, Wf = (float)g.W;
  for (long wk = wave0; wk < nwork; wk += nwaves) {
    const int grp = (int)(wk % g.dg);
    long r = wk / g.dg;
    const int t = (int)(r % taps);
    const long m = r / taps;
    const int wo = (int)(m % g.Wo);
    const int ho = (int)((m / g.Wo) % g.Ho);
    const int n = (int)(m / ((long)g.Wo * g.Ho));
    const int i = t / g.kw, j = t % g.kw;
    const int h_in = ho * g.stride - g.pad_t, w_in = wo * g.stride - g.pad_l;
    const bf16_t* op = offs + m * offc + (grp * taps + t) * 2;
    const float inv_h = (float)(h_in + i * g.dil) + bf2f(op[0]);
    const float inv_w = (float)(w_in + j * g.dil) + bf2f(op[1]);
    const int c = grp * 64 + lane;
    const float cg = bf2f(dS[(m * taps + t) * g.C + c]);
    const bf16_t* base = x + ((long)n * g.H * g.W) * g.C + c;
    // ---- dOffset (get_coordinate_weight)
    float s_h = 0.f, s_w = 0.f;
    if (!(inv_h < 0 || inv_w < 0 || inv_h >= Hf || inv_w >= Wf)) {
      float ih = inv_h, iw = inv_w;
      int h_low = (int)ih, w_low = (int)iw, h_high, w_high;
      if (h_low >= g.H - 1) { h_high = h_low = g.H - 1; ih = (float)h_low; } else h_high = h_low + 1;
      if (w_low >= g.W - 1) { w_high = w_low = g.W - 1; iw = (float)w_low; } else w_high = w_low + 1;
      const float vll = bf2f(base[((long)h_low * g.W + w_low) * g.C]), vlh = bf2f(base[((long)h_low * g.W + w_high) * g.C]);
      const float vhl = bf2f(base[((long)h_high * g.W + w_low) * g.C]), vhh = bf2f(base[((long)h_high * g.W + w_high) * g.C]);
      const float a_w = (float)(w_low + 1) - iw, b_w = iw - (float)w_low;
      const float a_h = (float)(h_low + 1) - ih, b_h = ih - (float)h_low;
      s_h = (-1.f * a_w * vll + -1.f * b_w * vlh + a_w * vhl + b_w * vhh) * cg;
      s_w = (-1.f * a_h * vll + a_h * vlh + -1.f * b_h * vhl + b_h * vhh) * cg;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_h += __shfl_xor(s_h, o, 64); s_w += __shfl_xor(s_w, o, 64); }
    if (lane == 0) {
      bf16_t* dp = doffs + m * offc + (grp * taps + t) * 2;
      dp[0] = f2bf(s_h);
      dp[1] = f2bf(s_w);
    }
    // ---- dX (get_gradient_weight), wave-uniform corner geometry
    if (!(inv_h < 0 || inv_h > Hf || inv_w < 0 || inv_w > Wf)) {
      float ah = fmaxf(inv_h, 0.f), aw = fmaxf(inv_w, 0.f);
      int hl = (int)ah, wl = (int)aw, hh, wh;
      const bool ch = hl >= g.H - 1, cw = wl >= g.W - 1;
      if (ch) { hh = hl = g.H - 1; ah = (float)hl; } else hh = hl + 1;
      if (cw) { wh = wl = g.W - 1; aw = (float)wl; } else wh = wl + 1;
      const int rh[4] = {hl, hl, hh, hh}, rw[4] = {wl, wh, wl, wh};
      const float wg[4] = {((float)(hl + 1) - ah) * ((float)(wl + 1) - aw), ((float)(hl + 1) - ah) * (aw + 1.f - (float)wh),
                           (ah + 1.f - (float)hh) * ((float)(wl + 1) - aw), (ah + 1.f - (float)hh) * (aw + 1.f - (float)wh)};
      const bool dup[4] = {false, cw, ch, ch || cw};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = !dup[k] && fabsf(inv_h - (float)rh[k]) < 1.f && fabsf(inv_w - (float)rw[k]) < 1.f && rh[k] >= 0 && rh[k] < g.H &&
                        rw[k] >= 0 && rw[k] < g.W;
        if (ok) atomicAdd(dx + (((long)n * g.H + rh[k]) * g.W + rw[k]) * g.C + c, wg[k] * cg);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Atomic-free backward for the shape the context modules use (3x3 taps, stride 1, C / deformable_group == 64).
// L2 float atomics run at ~1.4 TB/s on this part and the scatter issues 36 of them (256 B each) per output pixel and group, so the
// col2im is turned around: a wave owns one INPUT pixel (x 64 channels), enumerates the (output pixel, tap) pairs whose nominal sampling
// position lies within +-R pixels, evaluates get_gradient_weight for each in one lane, and accumulates the matching dS rows — a
// deterministic gather.  Corners further than R from their nominal position (offsets beyond ~R-1 px) keep the memory atomic
// (`far` below); the two sets are complementary by construction.  dOffset needs no atomics at all: one wave per (output pixel, group)
// has all nine taps' loads in flight and reduces the 18 sums with a transposing butterfly (29 shuffles instead of 108).

__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

struct CornerSet {
  int rh[4], rw[4];
  float wg[4];
  bool ok[4];
};
// get_gradient_weight (deform_conv.cu:130-173) + the guards of deformable_col2im_gpu_kernel (:311-326) for one sampling position
__device__ __forceinline__ CornerSet corner_set(float inv_h, float inv_w, int H, int W) {
  CornerSet cs;
  const bool in = !(inv_h < 0 || inv_h > (float)H || inv_w < 0 || inv_w > (float)W);
  float ah = fmaxf(inv_h, 0.f), aw = fmaxf(inv_w, 0.f);
  if (!in) { ah = 0.f; aw = 0.f; }
  int hl = (int)ah, wl = (int)aw, hh, wh;
  const bool ch = hl >= H - 1, cw = wl >= W - 1;
  if (ch) { hh = hl = H - 1; ah = (float)hl; } else hh = hl + 1;
  if (cw) { wh = wl = W - 1; aw = (float)wl; } else wh = wl + 1;
  cs.rh[0] = hl; cs.rh[1] = hl; cs.rh[2] = hh; cs.rh[3] = hh;
  cs.rw[0] = wl; cs.rw[1] = wh; cs.rw[2] = wl; cs.rw[3] = wh;
  cs.wg[0] = ((float)(hl + 1) - ah) * ((float)(wl + 1) - aw);
  cs.wg[1] = ((float)(hl + 1) - ah) * (aw + 1.f - (float)wh);
  cs.wg[2] = (ah + 1.f - (float)hh) * ((float)(wl + 1) - aw);
  cs.wg[3] = (ah + 1.f - (float)hh) * (aw + 1.f - (float)wh);
  const bool dup[4] = {false, cw, ch, ch || cw};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    cs.ok[k] = in && !dup[k] && fabsf(inv_h - (float)cs.rh[k]) < 1.f && fabsf(inv_w - (float)cs.rw[k]) < 1.f && cs.rh[k] >= 0 && cs.rh[k] < H &&
               cs.rw[k] >= 0 && cs.rw[k] < W;
  return cs;
}

// The weight corner_set() gives the corner that IS pixel (h, w), or 0: wg[k] and ok[k] factor into a row part and a column part
// (dup[1] = cw, dup[2] = ch, dup[3] = ch || cw), so the gather evaluates the two 1-D factors directly.
__device__ __forceinline__ float axis_factor(float inv, int target, int L) {
  float a = fmaxf(inv, 0.f);
  int lo = (int)a, hi;
  const bool clamp = lo >= L - 1;
  if (clamp) { hi = lo = L - 1; a = (float)lo; } else hi = lo + 1;
  const bool near = fabsf(inv - (float)target) < 1.f;
  float f = 0.f;
  if (target == lo) f = (float)(lo + 1) - a;                 // corners 0/1 (rows) or 0/2 (columns)
  else if (target == hi && !clamp) f = a + 1.f - (float)hi;  // the high corner, unless it duplicates the low one
  return near ? f : 0.f;
}
__device__ __forceinline__ float corner_weight_at(float inv_h, float inv_w, int h, int w, int H, int W) {
  if (inv_h < 0 || inv_h > (float)H || inv_w < 0 || inv_w > (float)W) return 0.f;
  return axis_factor(inv_h, h, H) * axis_factor(inv_w, w, W);
}

// The 18 sums of an item, spread over its 8 lanes -> lane l8 holds the (dh, dw) pair of tap (4*b0 + 2*b1 + b2) and stores it as one 32-bit
// word: a transposing butterfly (14 + 6 shuffles).  `dp` = the item's 18 bf16 of dOffset (an even element index: 32-bit aligned pairs).
__device__ __forceinline__ void doff_reduce_store(const float (&sv)[18], bool live, bf16_t* __restrict__ dp, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  // 16 of the 18 sums: after the step with partner distance d a lane keeps the half its bit selects
  float r8[8], r4[4], r2[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) r8[j] = (b0 ? sv[j + 8] : sv[j]) + __shfl_xor(b0 ? sv[j] : sv[j + 8], 1, 64);
#pragma unroll
  for (int j = 0; j < 4; ++j) r4[j] = (b1 ? r8[j + 4] : r8[j]) + __shfl_xor(b1 ? r8[j] : r8[j + 4], 2, 64);
#pragma unroll
  for (int j = 0; j < 2; ++j) r2[j] = (b2 ? r4[j + 2] : r4[j]) + __shfl_xor(b2 ? r4[j] : r4[j + 2], 4, 64);
  float e0 = sv[16], e1 = sv[17];
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) { e0 += __shfl_xor(e0, o, 64); e1 += __shfl_xor(e1, o, 64); }
  if (live) {
    *reinterpret_cast<unsigned*>(dp + (b0 ? 8 : 0) + (b1 ? 4 : 0) + (b2 ? 2 : 0)) = pack2bf(r2[0], r2[1]);
    if ((lane & 7) == 0) *reinterpret_cast<unsigned*>(dp + 16) = pack2bf(e0, e1);
  }
}

// Far corners: not reachable by the gather kernels' +-R enumeration.  A corner that corner_set() accepts sits floor(o) or
// floor(o) + 1 from the nominal position (under the high-edge clamp the only candidate one further away, H - 1 for a position of
// exactly H, fails the |position - corner| < 1 guard), so offsets in [-R, R) cannot produce one: the whole search is skipped then.
// `farm` = the item's taps with an offset outside [-R, R); uo / ud = the item's offsets and (lane's piece of the) dS rows.
template <int R>
__device__ __forceinline__ void doff_far_corners(unsigned farm, bool live, const bf16_t* __restrict__ uo, const bf16_t* __restrict__ ud,
                                                 float* __restrict__ far_dx, const DeformGeom& g, int n, int h_in, int w_in, int grp, int l8) {
  if (!__any(live && farm != 0)) return;
  if (!live) farm = 0;
  while (farm) {                                                        // (uniform inside an 8-lane item; rare)
    const int t = __builtin_ctz(farm);
    farm &= farm - 1;
    const unsigned oraw = *reinterpret_cast<const unsigned*>(uo + 2 * t);
    const int nh = h_in + (t / 3) * g.dil, nw = w_in + (t % 3) * g.dil;
    const CornerSet cs = corner_set((float)nh + bf2f((bf16_t)(oraw & 0xffffu)), (float)nw + bf2f((bf16_t)(oraw >> 16)), g.H, g.W);
    bool any_far = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) any_far = any_far || (cs.ok[k] && (abs(cs.rh[k] - nh) > R || abs(cs.rw[k] - nw) > R));
    if (!any_far) continue;
    float cg[8];
    unpack8(*reinterpret_cast<const uint4*>(ud + t * g.C), cg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool far = abs(cs.rh[k] - nh) > R || abs(cs.rw[k] - nw) > R;
      if (cs.ok[k] && far) {
        float* dst = far_dx + (((long)n * g.H + cs.rh[k]) * g.W + cs.rw[k]) * g.C + grp * 64 + l8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(dst + e, cs.wg[k] * cg[e]);
      }
    }
  }
}

// dOffset (+ the far corners' atomics into the fp32 side buffer): EIGHT LANES per (output pixel, group) — lane l8 owns 8 of the group's
// 64 channels, i.e. one 16-byte piece of every row it touches, so the 45 loads of an item are 16-byte lane accesses (the earlier
// wave-per-item form issued them as 2-byte ones: eight times the load instructions for the same bytes).  The sampling geometry is
// recomputed by each of the 8 lanes (cheaper than handing it around); the 18 sums are reduced over the 8 lanes with a transposing
// butterfly (14 + 6 shuffles) after which lane l8 holds the (dh, dw) pair of tap (4*b0 + 2*b1 + b2) and stores it as one 32-bit word.
template <int R>
__device__ __forceinline__ void deform_bwd_doff9_c64_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ offs,
                                                          const bf16_t* __restrict__ dS, float* __restrict__ far_dx,
                                                          bf16_t* __restrict__ doffs, const DeformGeom& g) {
  const int offc = g.dg * 18;
  const int lane = threadIdx.x & 63, l8 = lane & 7;
  const unsigned nwork = (unsigned)g.N * g.Ho * g.Wo * g.dg;            // (< 2^31: checked by the launcher)
  const float Hf = (float)g.H, Wf = (float)g.W;
  const float lim = (float)R;
  for (unsigned wbase = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8u; wbase < nwork; wbase += gridDim.x * (blockDim.x >> 6) * 8u) {
    const unsigned item = wbase + (lane >> 3);
    const bool live = item < nwork;
    const unsigned wk = live ? item : nwork - 1;
    const int grp = (int)(wk % (unsigned)g.dg);
    const unsigned m = wk / (unsigned)g.dg;
    const int wo = (int)(m % (unsigned)g.Wo);
    const unsigned mr = m / (unsigned)g.Wo;
    const int ho = (int)(mr % (unsigned)g.Ho), n = (int)(mr / (unsigned)g.Ho);
    const int h_in = ho - g.pad_t, w_in = wo - g.pad_l;
    const bf16_t* ub = x + ((long)n * g.H * g.W) * g.C + grp * 64 + l8 * 8;
    const bf16_t* ud = dS + (long)m * 9 * g.C + grp * 64 + l8 * 8;
    const bf16_t* uo = offs + (long)m * offc + grp * 18;
    float sv[18];
    unsigned farm = 0;                                                  // taps whose offset may reach a corner beyond +-R
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      uint4 v[3][4], cgr[3];
      float ca_w[3], cb_w[3], ca_h[3], cb_h[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = tb * 3 + u;
        const unsigned oraw = *reinterpret_cast<const unsigned*>(uo + 2 * t);
        const float off_h = bf2f((bf16_t)(oraw & 0xffffu)), off_w = bf2f((bf16_t)(oraw >> 16));
        if (off_h < -lim || off_h >= lim || off_w < -lim || off_w >= lim) farm |= 1u << t;
        const float inv_h = (float)(h_in + tb * g.dil) + off_h, inv_w = (float)(w_in + u * g.dil) + off_w;
        const bool in = !(inv_h < 0 || inv_w < 0 || inv_h >= Hf || inv_w >= Wf);
        float ih = in ? inv_h : 0.f, iw = in ? inv_w : 0.f;
        int hl = (int)ih, wl = (int)iw, hh, wh;
        if (hl >= g.H - 1) { hh = hl = g.H - 1; ih = (float)hl; } else hh = hl + 1;
        if (wl >= g.W - 1) { wh = wl = g.W - 1; iw = (float)wl; } else wh = wl + 1;
        ca_w[u] = in ? (float)(wl + 1) - iw : 0.f; cb_w[u] = in ? iw - (float)wl : 0.f;      // get_coordinate_weight (:177-221)
        ca_h[u] = in ? (float)(hl + 1) - ih : 0.f; cb_h[u] = in ? ih - (float)hl : 0.f;
        v[u][0] = *reinterpret_cast<const uint4*>(ub + (long)(hl * g.W + wl) * g.C);
        v[u][1] = *reinterpret_cast<const uint4*>(ub + (long)(hl * g.W + wh) * g.C);
        v[u][2] = *reinterpret_cast<const uint4*>(ub + (long)(hh * g.W + wl) * g.C);
        v[u][3] = *reinterpret_cast<const uint4*>(ub + (long)(hh * g.W + wh) * g.C);
        cgr[u] = *reinterpret_cast<const uint4*>(ud + t * g.C);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        float vll[8], vlh[8], vhl[8], vhh[8], cg[8];
        unpack8(v[u][0], vll); unpack8(v[u][1], vlh); unpack8(v[u][2], vhl); unpack8(v[u][3], vhh); unpack8(cgr[u], cg);
        const float a_w = ca_w[u], b_w = cb_w[u], a_h = ca_h[u], b_h = cb_h[u];
        float s_h = 0.f, s_w = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s_h += (-1.f * a_w * vll[e] + -1.f * b_w * vlh[e] + a_w * vhl[e] + b_w * vhh[e]) * cg[e];
          s_w += (-1.f * a_h * vll[e] + a_h * vlh[e] + -1.f * b_h * vhl[e] + b_h * vhh[e]) * cg[e];
        }
        sv[2 * (tb * 3 + u)] = s_h;
        sv[2 * (tb * 3 + u) + 1] = s_w;
      }
    }
    doff_reduce_store(sv, live, doffs + (long)m * offc + grp * 18, lane);
    doff_far_corners<R>(farm, live, uo, ud, far_dx, g, n, h_in, w_in, grp, l8);
  }
}

// dX: one wave per (input pixel, group), lane = candidate during the enumeration, lane = channel during the accumulation
// ONE launch for both gather windows (the window is picked from the statistics at run time): a form that does not run must not cost a
// launch of its own on a 65536-block grid (~150 us to retire empty; DAN-Deform has 12 of these calls per step).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void deform_bwd_doff9_c64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ offs,
                                                                   const bf16_t* __restrict__ dS, float* __restrict__ far_dx,
                                                                   bf16_t* __restrict__ doffs, DeformGeom g, BwdGate gate) {
  const int form = gate_form(gate);
  if (form == 0) deform_bwd_doff9_c64_body<1>(x, offs, dS, far_dx, doffs, g);
  else if (form == 1) deform_bwd_doff9_c64_body<2>(x, offs, dS, far_dx, doffs, g);
}

template <int R>
__device__ __forceinline__ void deform_bwd_dx_gather9_c64_body(const bf16_t* __restrict__ offs, const bf16_t* __restrict__ dS,
                                                               const float* __restrict__ far_dx, bf16_t* __restrict__ dx, const DeformGeom& g,
                                                               int accumulate, const bf16_t* __restrict__ relu_x) {
  constexpr int D = 2 * R + 1, NC = D * D * 9, ROUNDS = (NC + 63) / 64;
  const int offc = g.dg * 18;
  const int lane = threadIdx.x & 63;
  const long nwork = (long)g.N * g.H * g.W * g.dg;
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  for (long wk = wave0; wk < nwork; wk += nwaves) {
    const int grp = (int)(wk % g.dg);
    const long p = wk / g.dg;
    const int w = (int)(p % g.W);
    const int h = (int)((p / g.W) % g.H);
    const int n = (int)(p / ((long)g.W * g.H));
    const int c = grp * 64 + lane;
    unsigned oraw[ROUNDS];
    int row[ROUNDS], nhv[ROUNDS], nwv[ROUNDS];
    bool valid[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int cand = r * 64 + lane;
      const int t = cand % 9, pos = cand / 9;
      nhv[r] = h + pos / D - R;
      nwv[r] = w + pos % D - R;
      const int ho = nhv[r] + g.pad_t - (t / 3) * g.dil, wo = nwv[r] + g.pad_l - (t % 3) * g.dil;     // stride 1
      valid[r] = cand < NC && ho >= 0 && ho < g.Ho && wo >= 0 && wo < g.Wo;
      const long m = valid[r] ? ((long)n * g.Ho + ho) * g.Wo + wo : 0;
      row[r] = (int)(m * 9 + t);
      oraw[r] = *reinterpret_cast<const unsigned*>(offs + m * offc + (grp * 9 + t) * 2);
    }
    float wgt[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const float inv_h = (float)nhv[r] + bf2f((bf16_t)(oraw[r] & 0xffffu)), inv_w = (float)nwv[r] + bf2f((bf16_t)(oraw[r] >> 16));
      const float wv = corner_weight_at(inv_h, inv_w, h, w, g.H, g.W);
      wgt[r] = valid[r] ? wv : 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      unsigned long long mask = __ballot(wgt[r] != 0.f);
      while (mask) {                                                    // four dS rows in flight per trip
        float w4[4];
        int r4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            w4[q] = lane_f(wgt[r], b);
            r4[q] = __builtin_amdgcn_readlane(row[r], b);
          } else { w4[q] = 0.f; r4[q] = 0; }
        }
        bf16_t d4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d4[q] = (dS + (long)r4[q] * g.C + grp * 64)[lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += w4[q] * bf2f(d4[q]);
      }
    }
    const long o = p * g.C + c;
    float out = acc + far_dx[o];
    if (relu_x && !(bf2f(relu_x[o]) > 0.f)) out = 0.f;              // x is a ReLU output: its producer's ReLU backward, folded in here
    if (accumulate) out += bf2f(dx[o]);
    dx[o] = f2bf(out);
  }
}

// (8 waves per SIMD: the gather is two dependent global-load phases per item - offsets, then the dS rows they select - so it lives on
// occupancy; capping it at 64 VGPRs (4 spilled in the +-2 px body) measured 2.02 -> 1.96 ms at zero offsets, 3.20 -> 3.02 at sigma = 0.3 px
// (160 x 160 x 256, batch 16).  The opposite trade - prefetching the next item's offsets, 90 VGPRs, 5 waves - was 7-13 % SLOWER.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void deform_bwd_dx_gather9_c64_kernel(const bf16_t* __restrict__ offs, const bf16_t* __restrict__ dS,
                                                                        const float* __restrict__ far_dx, bf16_t* __restrict__ dx, DeformGeom g,
                                                                        int accumulate, BwdGate gate, const bf16_t* __restrict__ relu_x) {
  const int form = gate_form(gate);
  if (form == 0) deform_bwd_dx_gather9_c64_body<1>(offs, dS, far_dx, dx, g, accumulate, relu_x);
  else if (form == 1) deform_bwd_dx_gather9_c64_body<2>(offs, dS, far_dx, dx, g, accumulate, relu_x);
}

constexpr int DT_H = 8, DT_W = 16;

// dOffset in a +-R window (R = 1, 2), TILE form (round 5).  The item-per-8-lanes kernel above reads the four bilinear corners of all nine taps
// from L1 / L2: 36 sixteen-byte gathers of x per 9 of dS, and at 160 x 160 x 256 (batch 16) the cache pipes, not HBM, set its 0.77 ms.
// Here a 256-thread block owns an 8 x 16 tile of OUTPUT pixels of one group and stages the group's 64 channels of the x pixels any tap
// with an offset in [-R, R) can touch (tile +-(R + 1): 12 x 20 pixels, 30 KB for R = 1) in LDS once; the corner gathers become ds_read_b128.
// A tap whose corners leave the staged region (offset outside [-R, R): rare in the form the statistic picks) takes the global loads.  Arithmetic,
// reduction and the far corners' atomics are the item kernel's.  3 x 3, stride 1, dilation 1, C / deformable_group == 64.
template <int R>
__device__ __forceinline__ void deform_bwd_doff_tile_c64_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ offs,
                                                              const bf16_t* __restrict__ dS, float* __restrict__ far_dx,
                                                              bf16_t* __restrict__ doffs, const DeformGeom& g, int tiles_h, int tiles_w, uint4* sx) {
  constexpr int DT_RH = DT_H + 2 * (R + 1), DT_RW = DT_W + 2 * (R + 1);
  // XCD-aware order: workgroup i runs on XCD i % 8; each XCD walks a contiguous eighth of the (image, tile, group) items, so the four groups of
  // a tile (the same offset lines, neighbouring slices of the same x / dS rows) and the overlapping regions of neighbouring tiles meet in ONE L2
  const unsigned nitems = (unsigned)g.N * (unsigned)tiles_h * (unsigned)tiles_w * (unsigned)g.dg, per_xcd = (nitems + 7u) / 8u;
  unsigned b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (b >= nitems) return;                                               // (uniform; the grid is 8 * per_xcd)
  const int grp = (int)(b % (unsigned)g.dg); b /= (unsigned)g.dg;
  const int tw0 = (int)(b % (unsigned)tiles_w) * DT_W; b /= (unsigned)tiles_w;
  const int th0 = (int)(b % (unsigned)tiles_h) * DT_H;
  const int n = (int)(b / (unsigned)tiles_h);
  const int offc = g.dg * 18;
  const bf16_t* xn = x + ((long)n * g.H * g.W) * g.C + grp * 64;
  for (int idx = threadIdx.x; idx < DT_RH * DT_RW * 8; idx += 256) {
    const int px = idx >> 3, pc = idx & 7;
    const int hh = th0 - (R + 1) + px / DT_RW, ww = tw0 - (R + 1) + px % DT_RW;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (hh >= 0 && hh < g.H && ww >= 0 && ww < g.W) v = *reinterpret_cast<const uint4*>(xn + (long)(hh * g.W + ww) * g.C + pc * 8);
    sx[idx] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, l8 = lane & 7;
  const int slot = threadIdx.x >> 3;
  const int pw = slot % DT_W, ph0 = slot / DT_W;
  const float Hf = (float)g.H, Wf = (float)g.W;
#pragma unroll 1
  for (int bt = 0; bt < 4; ++bt) {
    const int ho_raw = th0 + 2 * bt + ph0, wo_raw = tw0 + pw;
    const bool live = ho_raw < g.Ho && wo_raw < g.Wo;
    const int ho = live ? ho_raw : g.Ho - 1, wo = live ? wo_raw : g.Wo - 1;
    const long m = ((long)n * g.Ho + ho) * g.Wo + wo;
    const int h_in = ho - g.pad_t, w_in = wo - g.pad_l;
    const bf16_t* ub = xn + l8 * 8;
    const bf16_t* ud = dS + m * 9 * g.C + grp * 64 + l8 * 8;
    const bf16_t* uo = offs + m * offc + grp * 18;
    unsigned farm = 0;                                                  // taps whose offset may reach a corner beyond +-R
    bf16_t* dp = doffs + m * offc + grp * 18;
    // A tap ROW at a time, the loop kept rolled: unrolled over all nine taps hipcc computes every tap's geometry up front (290 VGPRs).
    // The row's six sums are reduced over the item's 8 lanes by a transposing butterfly (4 + 2 + 1 shuffles): lane l8 ends with sum
    // 4*b0 + 2*b1 + b2 (6 and 7 are padding) and stores it as one 16-bit value.
#pragma unroll 1
    for (int tb = 0; tb < 3; ++tb) {
      uint4 cgr[3];
      unsigned orw[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) { cgr[u] = *reinterpret_cast<const uint4*>(ud + (tb * 3 + u) * g.C); orw[u] = *reinterpret_cast<const unsigned*>(uo + 2 * (tb * 3 + u)); }
      float sv[8];
      sv[6] = 0.f; sv[7] = 0.f;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const float off_h = bf2f((bf16_t)(orw[u] & 0xffffu)), off_w = bf2f((bf16_t)(orw[u] >> 16));
        if (off_h < -(float)R || off_h >= (float)R || off_w < -(float)R || off_w >= (float)R) farm |= 1u << (tb * 3 + u);
        const float inv_h = (float)(h_in + tb) + off_h, inv_w = (float)(w_in + u) + off_w;
        const bool in = !(inv_h < 0 || inv_w < 0 || inv_h >= Hf || inv_w >= Wf);
        float ih = in ? inv_h : 0.f, iw = in ? inv_w : 0.f;
        int hl = (int)ih, wl = (int)iw, hh, wh;
        if (hl >= g.H - 1) { hh = hl = g.H - 1; ih = (float)hl; } else hh = hl + 1;
        if (wl >= g.W - 1) { wh = wl = g.W - 1; iw = (float)wl; } else wh = wl + 1;
        const float a_w = in ? (float)(wl + 1) - iw : 0.f, b_w = in ? iw - (float)wl : 0.f;      // get_coordinate_weight (:177-221)
        const float a_h = in ? (float)(hl + 1) - ih : 0.f, b_h = in ? ih - (float)hl : 0.f;
        const int rl = hl - (th0 - (R + 1)), rh = hh - (th0 - (R + 1)), cl = wl - (tw0 - (R + 1)), ch = wh - (tw0 - (R + 1));
        const bool staged = rl >= 0 && rh < DT_RH && cl >= 0 && ch < DT_RW;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
        if (in && staged) {
          v0 = sx[(rl * DT_RW + cl) * 8 + l8]; v1 = sx[(rl * DT_RW + ch) * 8 + l8];
          v2 = sx[(rh * DT_RW + cl) * 8 + l8]; v3 = sx[(rh * DT_RW + ch) * 8 + l8];
        }
        if (__any(in && !staged)) {                                     // (an offset outside [-R, R): rare in this form)
          if (in && !staged) {
            v0 = *reinterpret_cast<const uint4*>(ub + (long)(hl * g.W + wl) * g.C); v1 = *reinterpret_cast<const uint4*>(ub + (long)(hl * g.W + wh) * g.C);
            v2 = *reinterpret_cast<const uint4*>(ub + (long)(hh * g.W + wl) * g.C); v3 = *reinterpret_cast<const uint4*>(ub + (long)(hh * g.W + wh) * g.C);
          }
        }
        // get_coordinate_weight's two sums, with the corner weights (the same for every channel) taken out of the channel sum: four dot
        // products of the packed dS piece with the packed corner pieces (v_dot2c: 16 instructions, nothing unpacked - the loop was
        // VALU-bound on 40 unpacks + 80 multiply-adds per tap), then s_h = a_w (D_hl - D_ll) + b_w (D_hh - D_lh), s_w likewise
        const unsigned cgw[4] = {cgr[u].x, cgr[u].y, cgr[u].z, cgr[u].w};
        const unsigned w0[4] = {v0.x, v0.y, v0.z, v0.w}, w1[4] = {v1.x, v1.y, v1.z, v1.w}, w2[4] = {v2.x, v2.y, v2.z, v2.w}, w3[4] = {v3.x, v3.y, v3.z, v3.w};
        float d_ll = 0.f, d_lh = 0.f, d_hl = 0.f, d_hh = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d_ll = dh_dot2(cgw[e], w0[e], d_ll); d_lh = dh_dot2(cgw[e], w1[e], d_lh);
          d_hl = dh_dot2(cgw[e], w2[e], d_hl); d_hh = dh_dot2(cgw[e], w3[e], d_hh);
        }
        sv[2 * u] = a_w * (d_hl - d_ll) + b_w * (d_hh - d_lh);
        sv[2 * u + 1] = a_h * (d_lh - d_ll) + b_h * (d_hh - d_hl);
      }
      const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
      float r4[4], r2[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) r4[j] = (b0 ? sv[j + 4] : sv[j]) + __shfl_xor(b0 ? sv[j] : sv[j + 4], 1, 64);
#pragma unroll
      for (int j = 0; j < 2; ++j) r2[j] = (b1 ? r4[j + 2] : r4[j]) + __shfl_xor(b1 ? r4[j] : r4[j + 2], 2, 64);
      const float r1 = (b2 ? r2[1] : r2[0]) + __shfl_xor(b2 ? r2[0] : r2[1], 4, 64);
      const int which = (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);
      if (live && which < 6) dp[tb * 6 + which] = f2bf(r1);
    }
    doff_far_corners<R>(farm, live, uo, ud, far_dx, g, n, h_in, w_in, grp, l8);
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void deform_bwd_doff_tile_c64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ offs,
                                                                       const bf16_t* __restrict__ dS, float* __restrict__ far_dx,
                                                                       bf16_t* __restrict__ doffs, DeformGeom g, BwdGate gate, int tiles_h, int tiles_w) {
  __shared__ uint4 sx[(DT_H + 6) * (DT_W + 6) * 8];                     // the +-2 window's region: 14 x 22 pixels x 128 B = 39 KB
  const int form = gate_form(gate);
  if (form == 0) deform_bwd_doff_tile_c64_body<1>(x, offs, dS, far_dx, doffs, g, tiles_h, tiles_w, sx);
  else if (form == 1) deform_bwd_doff_tile_c64_body<2>(x, offs, dS, far_dx, doffs, g, tiles_h, tiles_w, sx);
}

// dX in a +-R window (R = 1: 81 candidates per input pixel, R = 2: 225), TILE form (round 5).  The wave-per-pixel gather above spends its life on memory round trips: per input pixel and
// group it loads 81 candidates' offsets from 81 scattered 4-byte places, then the dS rows they select as 2-byte lane accesses (1.96 ms at
// 160 x 160 x 256, batch 16, zero offsets: 9 rows per pixel).  Here a 256-thread block owns an 8 x 16 tile of INPUT pixels of one group:
// the offsets of every (output pixel, tap) that can reach the tile (output pixels within +-(R + 1): 12 x 20 x 9 words for R = 1) are staged in LDS once,
// EIGHT lanes own a pixel (lane l8 = 8 of the group's 64 channels = one 16-byte piece of each dS row), the 81 candidates are weighed
// eight at a time (one per lane, offsets from LDS), and every non-zero one is handed to the pixel's eight lanes by two ds_bpermutes and
// accumulated from ONE 16-byte load per lane.  Same candidate set, same weights (corner_weight_at) and the same order (candidate
// index ascending) as the wave-per-pixel form: deterministic, no atomics; far corners as before (the fp32 side buffer, read only when
// the statistic says an offset left [-R, R) at all).  3 x 3, stride 1, dilation 1, C / deformable_group == 64.

template <int R>
__device__ __forceinline__ void deform_bwd_dx_tile_c64_body(const bf16_t* __restrict__ offs, const bf16_t* __restrict__ dS,
                                                            const float* __restrict__ far_dx, bf16_t* __restrict__ dx, const DeformGeom& g,
                                                            int accumulate, const BwdGate& gate, const bf16_t* __restrict__ relu_x, int tiles_h,
                                                            int tiles_w, unsigned* soff) {
  constexpr int DT_RH = DT_H + 2 * (R + 1), DT_RW = DT_W + 2 * (R + 1), D = 2 * R + 1, NC = D * D * 9, ROUNDS = (NC + 7) / 8;
  // XCD-aware order: workgroup i runs on XCD i % 8; each XCD walks a contiguous eighth of the (image, tile, group) items, so the four groups of
  // a tile (the same offset lines, neighbouring slices of the same x / dS rows) and the overlapping regions of neighbouring tiles meet in ONE L2
  const unsigned nitems = (unsigned)g.N * (unsigned)tiles_h * (unsigned)tiles_w * (unsigned)g.dg, per_xcd = (nitems + 7u) / 8u;
  unsigned b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (b >= nitems) return;                                               // (uniform; the grid is 8 * per_xcd)
  const int grp = (int)(b % (unsigned)g.dg); b /= (unsigned)g.dg;
  const int tw0 = (int)(b % (unsigned)tiles_w) * DT_W; b /= (unsigned)tiles_w;
  const int th0 = (int)(b % (unsigned)tiles_h) * DT_H;
  const int n = (int)(b / (unsigned)tiles_h);
  const int offc = g.dg * 18;
  const bool has_far = gate.stat[R == 1 ? 1 : 0] != 0u;                     // an offset outside [-R, R) exists at all
  {
    const unsigned far_away = (unsigned)f2bf(60000.f) * 0x10001u;            // a position no pixel is near: weight 0
    for (int idx = threadIdx.x; idx < DT_RH * DT_RW * 9; idx += 256) {
      const int t = idx % 9, r = idx / 9;
      const int ho = th0 - (R + 1) + r / DT_RW, wo = tw0 - (R + 1) + r % DT_RW;
      unsigned v = far_away;
      if (ho >= 0 && ho < g.Ho && wo >= 0 && wo < g.Wo)
        v = *reinterpret_cast<const unsigned*>(offs + (((long)n * g.Ho + ho) * g.Wo + wo) * offc + (grp * 9 + t) * 2);
      soff[idx] = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, l8 = lane & 7, gbase = lane & 56;
  const int slot = threadIdx.x >> 3;                                         // 0..31: pixel (2 * bt + slot / 16, slot % 16) of the tile in batch bt
  const int pw = slot % DT_W, ph0 = slot / DT_W;
  const int w = tw0 + pw;
  const bf16_t* dsg = dS + grp * 64 + l8 * 8;
  // (batch outermost, one pixel's eight sums live at a time: 8 waves per SIMD — the loop is two dependent round trips per candidate round)
#pragma unroll 1
  for (int bt = 0; bt < 4; ++bt) {
    const int ph = 2 * bt + ph0, h = th0 + ph;
    const bool pvalid = h < g.H && w < g.W;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
      const int cand = r * 8 + l8;
      const bool live = cand < NC && pvalid;
      const int t = cand % 9, pos = cand < NC ? cand / 9 : (D * D) / 2;
      const int dh = pos / D - R, dw = pos % D - R;
      const int ti = t / 3, tj = t % 3;
      // output pixel of the candidate: (nominal position) + pad - tap, pad = 1: row h + dh + 1 - ti, staged at row index (that) - (th0 - R - 1)
      const int nh = h + dh, nw = w + dw;
      const unsigned oraw = soff[((ph + dh + R + 2 - ti) * DT_RW + (pw + dw + R + 2 - tj)) * 9 + t];
      const float inv_h = (float)nh + bf2f((bf16_t)(oraw & 0xffffu)), inv_w = (float)nw + bf2f((bf16_t)(oraw >> 16));
      float wv = corner_weight_at(inv_h, inv_w, h, w, g.H, g.W);
      if (!live) wv = 0.f;
      const int row = (((n * g.Ho + (nh + 1 - ti)) * g.Wo) + (nw + 1 - tj)) * 9 + t;       // (used only where wv != 0: a valid output pixel)
      unsigned m8 = (unsigned)(__ballot(wv != 0.f) >> gbase) & 0xffu;
      while (__any(m8 != 0u)) {                                              // four dS rows in flight per trip and pixel
        float w4[4];
        int r4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool has = m8 != 0u;
          const int src = gbase + (has ? __builtin_ctz(m8) : 0);
          m8 &= m8 - 1u;
          const float wq = __shfl(wv, src, 64);
          const int rq = __shfl(row, src, 64);
          w4[q] = has ? wq : 0.f;
          r4[q] = has ? rq : 0;
        }
        uint4 d4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d4[q] = *reinterpret_cast<const uint4*>(dsg + (long)r4[q] * g.C);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float f[8];
          unpack8(d4[q], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w4[q] * f[e];
        }
      }
    }
    if (!pvalid) continue;
    const long o = (((long)n * g.H + h) * g.W + w) * g.C + grp * 64 + l8 * 8;
    if (has_far) {
      const float4 fa = *reinterpret_cast<const float4*>(far_dx + o), fb = *reinterpret_cast<const float4*>(far_dx + o + 4);
      acc[0] += fa.x; acc[1] += fa.y; acc[2] += fa.z; acc[3] += fa.w; acc[4] += fb.x; acc[5] += fb.y; acc[6] += fb.z; acc[7] += fb.w;
    }
    if (relu_x) {                                                            // x is a ReLU output: its producer's ReLU backward, folded in here
      float m[8];
      unpack8(*reinterpret_cast<const uint4*>(relu_x + o), m);
#pragma unroll
      for (int e = 0; e < 8; ++e) if (!(m[e] > 0.f)) acc[e] = 0.f;
    }
    if (accumulate) {
      float old[8];
      unpack8(*reinterpret_cast<const uint4*>(dx + o), old);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += old[e];
    }
    *reinterpret_cast<uint4*>(dx + o) = pack8(acc);
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void deform_bwd_dx_tile_c64_kernel(const bf16_t* __restrict__ offs, const bf16_t* __restrict__ dS,
                                                                     const float* __restrict__ far_dx, bf16_t* __restrict__ dx, DeformGeom g,
                                                                     int accumulate, BwdGate gate, const bf16_t* __restrict__ relu_x, int tiles_h,
                                                                     int tiles_w) {
  __shared__ unsigned soff[(DT_H + 6) * (DT_W + 6) * 9];                     // the +-2 window's region: 14 x 22 output pixels x 9 taps = 11 KB
  const int form = gate_form(gate);
  if (form == 0) deform_bwd_dx_tile_c64_body<1>(offs, dS, far_dx, dx, g, accumulate, gate, relu_x, tiles_h, tiles_w, soff);
  else if (form == 1) deform_bwd_dx_tile_c64_body<2>(offs, dS, far_dx, dx, g, accumulate, gate, relu_x, tiles_h, tiles_w, soff);
}

// The fp32 side buffer is needed only when a corner can fall outside the gather window or the scatter form runs: both imply an offset
// outside [-1, 1) (stat[1] != 0) unless a form is forced.  420 MB of zero stores per call at 160 x 160 x 256, batch 16, otherwise.
__global__ void zero_if_far_kernel(uint4* __restrict__ p16, long n16, const unsigned* __restrict__ stat, int force) {
  if (!(stat[1] != 0u || force == 1 || force == 2)) return;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p16[i] = z;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n8, int accumulate, BwdGate gate,
                                   const bf16_t* __restrict__ relu_x) {
  if (gate.stat && !gate_runs(gate, 2)) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(src + i * 8), b = *reinterpret_cast<const float4*>(src + i * 8 + 4);
    float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (relu_x) {
      float m[8];
      unpack8(*reinterpret_cast<const uint4*>(relu_x + i * 8), m);
#pragma unroll
      for (int e = 0; e < 8; ++e) if (!(m[e] > 0.f)) f[e] = 0.f;
    }
    if (accumulate) {
      float o[8];
      unpack8(*reinterpret_cast<const uint4*>(dst + i * 8), o);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += o[e];
    }
    *reinterpret_cast<uint4*>(dst + i * 8) = pack8(f);
  }
}

// offsets bf16 [pairs][2] -> stat[0] / stat[1] += number of pairs with dh or dw outside [-2, 2) / [-1, 1)
__global__ __launch_bounds__(256) void deform_far_stat_kernel(const bf16_t* __restrict__ offs, long pairs, unsigned* __restrict__ stat) {
  unsigned c2 = 0, c1 = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
    const unsigned w = reinterpret_cast<const unsigned*>(offs)[i];
    const float dh = bf2f((bf16_t)(w & 0xffffu)), dw = bf2f((bf16_t)(w >> 16));
    c2 += (!(dh >= -2.f && dh < 2.f) || !(dw >= -2.f && dw < 2.f)) ? 1u : 0u;
    c1 += (!(dh >= -1.f && dh < 1.f) || !(dw >= -1.f && dw < 1.f)) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { c2 += __shfl_xor(c2, o, 64); c1 += __shfl_xor(c1, o, 64); }
  __shared__ unsigned part[2][4];                                        // one atomic per block and word (8192 waves queued on one address: 100 us)
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = c2; part[1][threadIdx.x >> 6] = c1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const unsigned v = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
    if (v) atomicAdd(stat + threadIdx.x, v);
  }
}

int make_geom(DeformGeom* g, int N, int H, int W, int C, int kh, int kw, int stride, int dil, int dg, const char* what) {
  DH_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && dg > 0, DANHIP_EINVAL, "%s: non-positive dims", what);
  DH_REQUIRE(C % dg == 0 && (C / dg) % 8 == 0, DANHIP_EINVAL, "%s: C/deformable_group must be a multiple of 8", what);
  const int cpg8 = C / dg / 8;
  DH_REQUIRE((cpg8 & (cpg8 - 1)) == 0 && cpg8 <= 32, DANHIP_EINVAL, "%s: C/deformable_group/8 must be a power of two <= 32", what);
  g->N = N; g->H = H; g->W = W; g->C = C; g->kh = kh; g->kw = kw; g->stride = stride; g->dil = dil; g->dg = dg;
  g->Ho = (H + stride - 1) / stride;
  g->Wo = (W + stride - 1) / stride;
  // SAME pad_before from the UNDILATED kernel (deform_conv.cc:473-479)
  int th = (g->Ho - 1) * stride + kh - H; if (th < 0) th = 0;
  int tw = (g->Wo - 1) * stride + kw - W; if (tw < 0) tw = 0;
  g->pad_t = th / 2; g->pad_l = tw / 2;
  DH_REQUIRE((int64_t)N * g->Ho * g->Wo * kh * kw * C < (1ll << 40), DANHIP_EINVAL, "%s: tensor too large", what);
  return DANHIP_OK;
}

inline int grid_for(long total, int block = 256, int cap = 16384) {
  long b = (total + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

/* S [N*Ho*Wo, kh*kw*C] bf16 = deformable im2col of x [N,H,W,C] under offsets [N,Ho,Wo,dg*2*kh*kw] (both bf16). */
extern "C" int danhip_deform_sample_fwd(const uint16_t* x, const uint16_t* offsets, uint16_t* S, int32_t N, int32_t H, int32_t W, int32_t C,
                                        int32_t kh, int32_t kw, int32_t stride, int32_t dilation, int32_t deformable_group, void* stream) {
  DH_REQUIRE(x && offsets && S, DANHIP_EINVAL, "deform_sample_fwd: null pointer");
  DeformGeom g;
  int rc = make_geom(&g, N, H, W, C, kh, kw, stride, dilation, deformable_group, "deform_sample_fwd");
  if (rc) return rc;
  const int segs = (g.Wo + SAMPLE_PXB - 1) / SAMPLE_PXB, cch = C / 8;
  const long blocks = (long)N * g.Ho * segs;
  DH_REQUIRE(blocks < (1l << 31), DANHIP_EINVAL, "deform_sample_fwd: %ld output row segments", blocks);
  int shift = -1;
  if ((cch & (cch - 1)) == 0) { shift = 0; while ((1 << shift) < cch) ++shift; }
  hipLaunchKernelGGL(deform_sample_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, offsets, S, g, segs, shift);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" size_t danhip_deform_sample_bwd_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  return ((size_t)N * H * W * C + 64) * sizeof(float);
}

/* dS [N*Ho*Wo, kh*kw*C] -> d_offsets bf16 [N,Ho,Wo,dg*2*kh*kw] (overwritten) and dx bf16 [N,H,W,C] (=|+= if accumulate).
 * workspace: N*H*W*C + 64 floats (fp32 scatter target + the far-corner statistic), zeroed inside. */
static int deform_sample_bwd_impl(const uint16_t* x, const uint16_t* offsets, const uint16_t* dS, uint16_t* dx, uint16_t* d_offsets,
                                  int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride, int32_t dilation,
                                  int32_t deformable_group, int accumulate, int relu_x, float* workspace, size_t workspace_bytes, void* stream) {
  const bf16_t* rx = relu_x ? x : nullptr;              // dx *= (x > 0): x is a ReLU output whose producer takes dx as delivered
  DH_REQUIRE(x && offsets && dS && dx && d_offsets && workspace, DANHIP_EINVAL, "deform_sample_bwd: null pointer");
  DH_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && workspace_bytes >= danhip_deform_sample_bwd_workspace_bytes(N, H, W, C), DANHIP_EWORKSPACE,
             "deform_sample_bwd: workspace of %zu bytes, needs %zu (N*H*W*C + 64 floats)", workspace_bytes,
             danhip_deform_sample_bwd_workspace_bytes(N, H, W, C));
  DeformGeom g;
  int rc = make_geom(&g, N, H, W, C, kh, kw, stride, dilation, deformable_group, "deform_sample_bwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long nx = (long)N * H * W * C;
  unsigned* stat = reinterpret_cast<unsigned*>(workspace + nx);
  const bool gathered = C / deformable_group == 64 && stride == 1 && kh == 3 && kw == 3 && (long)N * g.Ho * g.Wo * 9 < (1l << 31) && deformable_group <= 9;
  // fp32 scatter target + 64 words of statistics behind it: zeroed together — or, for the gather forms, the statistics now and the
  // target by zero_if_far_kernel once they say it will be used at all
  const int tiles_h = (H + DT_H - 1) / DT_H, tiles_w = (W + DT_W - 1) / DT_W;
  const long tile_blocks = (long)N * tiles_h * tiles_w * deformable_group;
  // both gather windows as the tile kernels (dilation 1); other dilations keep the item / wave-per-pixel kernels (which read the side buffer always)
  const bool tiled = gathered && dilation == 1 && g.pad_t == 1 && g.pad_l == 1 && tile_blocks < (1l << 31) - 8 && !danhip_option("deform_dx_untiled");
  const bool lazy_zero = tiled && (nx * 4) % 16 == 0;
  { const int zrc = lazy_zero ? danhip_zero_async(stat, sizeof(unsigned) * 64, s) : danhip_zero_async(workspace, sizeof(float) * (nx + 64), s); if (zrc) return zrc; }
  if (gathered) {
    const long nd = (long)N * g.Ho * g.Wo * deformable_group, ng = (long)N * H * W * deformable_group;
    const long pairs = (long)N * g.Ho * g.Wo * deformable_group * 9;
    // A corner lies floor(o) or floor(o) + 1 from the tap's nominal position, so it is outside the +-R window iff o is outside [-R, R).
    // Scatter form beyond 15 % of the taps outside [-2, 2) (160 x 160 x 256, batch 16, offsets N(0, s): s = 0.5 / 1.0 px -> 0 % / 9 % ->
    // 3.4 / 7.7 ms in the +-2 gather form; s = 2.0 -> 53 % -> 43 ms; the scatter form: 12.0 ms whatever the offsets); the +-1 window
    // (81 candidates per input pixel and group instead of 225) while fewer than 1 / 128 of the taps lie outside [-1, 1).
    BwdGate gate{stat, (unsigned)(pairs / 20 * 3), (unsigned)(pairs / 128), danhip_option("deform_bwd_form")};
    hipLaunchKernelGGL(deform_far_stat_kernel, dim3(grid_for(pairs, 256, 1024)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(offsets), pairs, stat);
    if (lazy_zero)
      hipLaunchKernelGGL(zero_if_far_kernel, dim3(grid_for(nx / 4, 256, 2048)), dim3(256), 0, s, reinterpret_cast<uint4*>(workspace), nx / 4, stat, gate.force);
    if (tiled) {
      const unsigned tgrid = (unsigned)((tile_blocks + 7) / 8 * 8);       // 8 XCDs x their share of the items (the kernels map blockIdx -> item)
      hipLaunchKernelGGL(deform_bwd_doff_tile_c64_kernel, dim3(tgrid), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(x),
                         reinterpret_cast<const bf16_t*>(offsets), reinterpret_cast<const bf16_t*>(dS), workspace, reinterpret_cast<bf16_t*>(d_offsets), g, gate,
                         tiles_h, tiles_w);
      hipLaunchKernelGGL(deform_bwd_dx_tile_c64_kernel, dim3(tgrid), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(offsets),
                         reinterpret_cast<const bf16_t*>(dS), workspace, reinterpret_cast<bf16_t*>(dx), g, accumulate, gate, rx, tiles_h, tiles_w);
    } else {
      const dim3 gd(grid_for((nd + 31) / 32 * 256, 256, 65536)), gg(grid_for((ng + 3) / 4 * 256, 256, 65536));
      hipLaunchKernelGGL(deform_bwd_doff9_c64_kernel, gd, dim3(256), 0, s, x, offsets, dS, workspace, d_offsets, g, gate);
      hipLaunchKernelGGL(deform_bwd_dx_gather9_c64_kernel, gg, dim3(256), 0, s, offsets, dS, workspace, dx, g, accumulate, gate, rx);
    }
    // the scatter form's two kernels on small grids (grid-stride loops): empty ~10 us each when a gather form runs
    const long nwork = (long)N * g.Ho * g.Wo * kh * kw * deformable_group;
    hipLaunchKernelGGL(deform_sample_bwd_c64_kernel, dim3(grid_for((nwork + 3) / 4 * 256, 256, 2048)), dim3(256), 0, s, x, offsets, dS, workspace, d_offsets, g,
                       gate);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(nx / 8, 256, 2048)), dim3(256), 0, s, workspace, dx, nx / 8, accumulate, gate, rx);
    DH_LAUNCH_CHECK();
    return DANHIP_OK;
  }
  const BwdGate none{nullptr, 0u, 0u, 0};
  if (C / deformable_group == 64) {
    const long nwork = (long)N * g.Ho * g.Wo * kh * kw * deformable_group;
    hipLaunchKernelGGL(deform_sample_bwd_c64_kernel, dim3(grid_for((nwork + 3) / 4 * 256, 256, 65536)), dim3(256), 0, s, x, offsets, dS, workspace, d_offsets, g,
                       none);
  } else {
    const long total = (long)N * g.Ho * g.Wo * kh * kw * (C / 8);
    hipLaunchKernelGGL(deform_sample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, offsets, dS, workspace, d_offsets, g);
  }
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(nx / 8)), dim3(256), 0, s, workspace, dx, nx / 8, accumulate, none, rx);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_deform_sample_bwd(const uint16_t* x, const uint16_t* offsets, const uint16_t* dS, uint16_t* dx, uint16_t* d_offsets,
                                        int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride, int32_t dilation,
                                        int32_t deformable_group, int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  return deform_sample_bwd_impl(x, offsets, dS, dx, d_offsets, N, H, W, C, kh, kw, stride, dilation, deformable_group, accumulate, 0, workspace,
                                workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// DeformConvOp / DeformConvBackpropOp as single entry points (cpp/Deform/deform_conv.cc:392-535, :635-771): the same
// orchestration as the reference's Compute() — im2col, GEMM; and for the backward GEMM^T, col2im_coord, col2im,
// re-im2col, filter GEMM — with the im2col buffer living in the caller's workspace only for the duration of the call
// (the reference allocates it with allocate_temp per call, :497-503) and all samples processed in one batch.
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t danhip_deform_conv_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride,
                                                     int backward) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return 0;
  const size_t Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const size_t col = align256((size_t)N * Ho * Wo * kh * kw * C * sizeof(uint16_t));
  return backward ? 2 * col + align256(((size_t)N * H * W * C + 64) * sizeof(float)) : col;
}

static int deform_gemm_desc(danhip_conv_desc* d, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride) {
  d->N = N; d->Ho = (H + stride - 1) / stride; d->Wo = (W + stride - 1) / stride;
  d->H = d->Ho; d->W = d->Wo; d->Cin = kh * kw * C; d->Cout = Cout; d->kh = 1; d->kw = 1; d->stride = 1;
  return 0;
}

bool danhip_deform_fused_eligible(int N, int H, int W, int C, int Cout, int kh, int kw, int stride, int dg);
int danhip_launch_deform_fused_fwd(const uint16_t* x, const uint16_t* wf_packed, int kpad, const float* bias, const uint16_t* offsets, uint16_t* y,
                                   uint16_t* col, int N, int H, int W, int C, int Cout, int stride, int dil, int dg, int relu, hipStream_t s);

extern "C" int danhip_deform_conv_fused(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                        int32_t deformable_group) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || stride <= 0 || deformable_group <= 0) return 0;
  return danhip_deform_fused_eligible(N, H, W, C, Cout, kh, kw, stride, deformable_group) ? 1 : 0;
}

extern "C" int danhip_deform_conv_fwd(const uint16_t* x, const uint16_t* wf_packed, const float* bias, const uint16_t* offsets, uint16_t* y,
                                      int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                      int32_t dilation, int32_t deformable_group, int relu, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  DH_REQUIRE(x && wf_packed && offsets && y, DANHIP_EINVAL, "deform_conv_fwd: null pointer");
  if (danhip_deform_conv_fused(N, H, W, C, Cout, kh, kw, stride, deformable_group)) {
    // one kernel: sampling feeds the GEMM through LDS; the column buffer is written only if the caller hands a workspace for it
    DeformGeom g;
    int rc = make_geom(&g, N, H, W, C, kh, kw, stride, dilation, deformable_group, "deform_conv_fwd");
    if (rc) return rc;
    DH_REQUIRE(!workspace || workspace_bytes >= danhip_deform_conv_workspace_bytes(N, H, W, C, kh, kw, stride, 0), DANHIP_EWORKSPACE,
               "deform_conv_fwd: workspace too small for the column buffer");
    const int kpad = (kh * kw * C + 63) / 64 * 64;
    return danhip_launch_deform_fused_fwd(x, wf_packed, kpad, bias, offsets, y, (uint16_t*)workspace, N, H, W, C, Cout, stride, dilation,
                                          deformable_group, relu, (hipStream_t)stream);
  }
  DH_REQUIRE(workspace, DANHIP_EINVAL, "deform_conv_fwd: this shape needs the column-buffer workspace");
  DH_REQUIRE(workspace_bytes >= danhip_deform_conv_workspace_bytes(N, H, W, C, kh, kw, stride, 0) && workspace_bytes > 0, DANHIP_EWORKSPACE,
             "deform_conv_fwd: workspace too small");
  uint16_t* col = (uint16_t*)workspace;
  int rc = danhip_deform_sample_fwd(x, offsets, col, N, H, W, C, kh, kw, stride, dilation, deformable_group, stream);
  if (rc) return rc;
  danhip_conv_desc d;
  deform_gemm_desc(&d, N, H, W, C, Cout, kh, kw, stride);
  return danhip_conv2d_fwd(&d, col, wf_packed, bias, y, DANHIP_BF16, relu, nullptr, stream);
}

// `col_saved`: the im2col buffer danhip_deform_conv_fwd left in ITS workspace (same x / offsets), kept alive by the caller — the backward
// then skips the reference's re-im2col (:744-748; 1.9 GB rewritten per call at 160x160x256, batch 16).  NULL: re-sample as the reference.
static int deform_conv_bwd_impl(const uint16_t* x, const uint16_t* wb_packed, const uint16_t* offsets, const uint16_t* dy,
                                const uint16_t* col_saved, uint16_t* dx, uint16_t* d_offsets, float* dw, float* db, int32_t N,
                                int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                int32_t dilation, int32_t deformable_group, int accumulate_dx, int relu_x, void* workspace,
                                size_t workspace_bytes, void* stream) {
  DH_REQUIRE(x && wb_packed && offsets && dy && dx && d_offsets && dw && workspace, DANHIP_EINVAL, "deform_conv_bwd: null pointer");
  const size_t need = danhip_deform_conv_workspace_bytes(N, H, W, C, kh, kw, stride, 1);
  DH_REQUIRE(workspace_bytes >= need && need > 0, DANHIP_EWORKSPACE, "deform_conv_bwd: workspace too small");
  const size_t colb = danhip_deform_conv_workspace_bytes(N, H, W, C, kh, kw, stride, 0);
  uint16_t* col = (uint16_t*)workspace;
  uint16_t* dcol = (uint16_t*)((char*)workspace + colb);
  float* scatter = (float*)((char*)workspace + 2 * colb);
  danhip_conv_desc d;
  deform_gemm_desc(&d, N, H, W, C, Cout, kh, kw, stride);
  int rc = danhip_conv2d_bwd_data(&d, dy, wb_packed, nullptr, dcol, 0, stream);                       // col gradient = W^T dOut (:700-712)
  if (rc) return rc;
  rc = deform_sample_bwd_impl(x, offsets, dcol, dx, d_offsets, N, H, W, C, kh, kw, stride, dilation, deformable_group, accumulate_dx, relu_x, scatter,
                              workspace_bytes - 2 * colb, stream);                                                                // col2im_coord + col2im (:716-741)
  if (rc) return rc;
  if (!col_saved) {
    rc = danhip_deform_sample_fwd(x, offsets, col, N, H, W, C, kh, kw, stride, dilation, deformable_group, stream);   // re-im2col (:744-748)
    if (rc) return rc;
  }
  return danhip_conv2d_bwd_weight(&d, col_saved ? col_saved : col, dy, dw, db, kh * kw * C, stream);  // dW += dOut col^T (:750-768)
}

extern "C" int danhip_deform_conv_bwd_with_col(const uint16_t* x, const uint16_t* wb_packed, const uint16_t* offsets, const uint16_t* dy,
                                               const uint16_t* col_saved, uint16_t* dx, uint16_t* d_offsets, float* dw, float* db, int32_t N,
                                               int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                               int32_t dilation, int32_t deformable_group, int accumulate_dx, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  return deform_conv_bwd_impl(x, wb_packed, offsets, dy, col_saved, dx, d_offsets, dw, db, N, H, W, C, Cout, kh, kw, stride, dilation, deformable_group,
                              accumulate_dx, 0, workspace, workspace_bytes, stream);
}

// The same with the input gradient DELIVERED the way the convolutions hand theirs over (dan_amd.ops.GradSlot): relu_x != 0 multiplies it by
// (x > 0) - x is then a ReLU output and its producer takes the gradient as final - before accumulate_dx adds what dx already holds.
extern "C" int danhip_deform_conv_bwd_deliver(const uint16_t* x, const uint16_t* wb_packed, const uint16_t* offsets, const uint16_t* dy,
                                              const uint16_t* col_saved, uint16_t* dx, uint16_t* d_offsets, float* dw, float* db, int32_t N,
                                              int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                              int32_t dilation, int32_t deformable_group, int accumulate_dx, int relu_x, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  return deform_conv_bwd_impl(x, wb_packed, offsets, dy, col_saved, dx, d_offsets, dw, db, N, H, W, C, Cout, kh, kw, stride, dilation, deformable_group,
                              accumulate_dx, relu_x, workspace, workspace_bytes, stream);
}

extern "C" int danhip_deform_conv_bwd(const uint16_t* x, const uint16_t* wb_packed, const uint16_t* offsets, const uint16_t* dy, uint16_t* dx,
                                      uint16_t* d_offsets, float* dw, float* db, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout,
                                      int32_t kh, int32_t kw, int32_t stride, int32_t dilation, int32_t deformable_group, int accumulate_dx,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return danhip_deform_conv_bwd_with_col(x, wb_packed, offsets, dy, nullptr, dx, d_offsets, dw, db, N, H, W, C, Cout, kh, kw, stride, dilation,
                                         deformable_group, accumulate_dx, workspace, workspace_bytes, stream);
}
