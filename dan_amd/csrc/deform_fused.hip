// DeformConvOp forward as ONE kernel: the deformable im2col (cpp/Deform/deform_conv.cu:229-275 + deformable_im2col_bilinear :91-126) feeds
// the GEMM (deform_conv.cc:505-530) through LDS instead of through a [M, kh*kw*C] column buffer in HBM — SURVEY §8 row a16.
//
// For the shape the context modules use (C / deformable_group == 64) a 64-channel K-step of the GEMM is exactly one (tap, deformable
// group) pair, so every pixel of the tile has ONE sampling position per K-step:
//   * a workgroup of 4 x FUSED_BM / 64 waves owns FUSED_BM output pixels x all Cout (<= 256) channels; per K-step a thread
//     reads the tap's offset pair, gathers 16 channels of the four bilinear corners (eight 16-byte loads, issued one K-step ahead),
//     blends them in fp32, rounds to the 16-bit activation type — the same values the column buffer would hold — and writes them into the
//     [128 px][64 ch] A tile in LDS (XOR-swizzled 16-byte pieces);
//   * the weight tile [Cout][64 k] of the step arrives by LDS-DMA (inline asm, so the compiler does not drain it before the fragment
//     reads of the CURRENT step);
//   * eight waves (2 x 4) run v_mfma_f32_16x16x32 on 64 px x Cout/4 wave tiles, weights = A operand (a lane owns 4 consecutive output
//     channels of one pixel), fp32 accumulators across the 9 * dg K-steps, bias / ReLU epilogue.
// `col` (optional): the rounded samples are ALSO stored as the column buffer, for a caller that keeps it for the filter gradient
// (dan_amd/ops.py KEEP_DEFORM_COL); inference passes NULL and the 9x activation-sized buffer never exists.
#include "common.h"

// 64 pixels per workgroup (256 threads, 80 KB of LDS at Cout = 256): TWO workgroups per CU, so one gathers while the other multiplies --
// 1.105 -> 1.045 ms at 160 x 160 x 256, offsets N(0, 2 px), against the 128-pixel / 512-thread form (one resident workgroup, 96 KB).
#ifndef FUSED_BM
#define FUSED_BM 64
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned df_u32x4;

__device__ __forceinline__ void df_dma16(df_u32x4 rsrc, unsigned voff, unsigned lds_addr) {      // voff out of range: zeros land in LDS
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void df_unpack8(const uint4& u, float* f) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack2bf(w[i], f[2 * i], f[2 * i + 1]);
}

struct FusedArgs {
  const bf16_t* x;        // [N,H,W,C]
  const bf16_t* offs;     // [N,Ho,Wo,dg*18]
  const bf16_t* w;        // packed [Cout_pad][Kpad], k = tap*C + c
  const float* bias;      // [Cout] or null
  bf16_t* y;              // [N,Ho,Wo,Cout]
  bf16_t* col;            // [N*Ho*Wo, 9*C] or null
  int N, H, W, C, Ho, Wo, Cout, Kpad, dg, stride, dil, pad_t, pad_l, relu;
  long M;
};

template <int BN, int BM>
__global__ __launch_bounds__(BM * 4, 2) void deform_fused_fwd_kernel(const FusedArgs a) {
  constexpr int NW = BM / 16;                               // waves: (BM / 64) x 4
  constexpr int NPT = 4, NCT = BN / 64;                     // wave tile 64 px x BN/4 co
  constexpr int ABYTES = BM * 128, WBYTES = BN * 128;       // one stage each
  constexpr int WPW = BN / 8 / NW;                          // weight DMA pieces (1 KiB = 8 rows) per wave and K-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS map: [A stage 0][A stage 1][W stage 0][W stage 1]
  const int tid = threadIdx.x, lane = tid & 63;
  // XCD-aware tile order (round 5): workgroup i runs on XCD i % 8, so with tile = blockIdx the 64 workgroups resident on an XCD were spread
  // over 512 consecutive tiles (205 image rows, 17 MB of x: its 4 MiB L2 kept nothing and the corner gathers fetched 4.9 GB per launch
  // from the fabric for a 210 MB input - rocprofv3 FETCH_SIZE).  Each XCD now walks its own contiguous eighth of the tiles: its resident
  // workgroups cover ~26 rows (2 MB) and the nine taps' gathers of a row hit L2.
  const unsigned per_xcd = ((unsigned)((a.M + BM - 1) / BM) + 7u) / 8u;
  const long tile = (long)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (tile * BM >= a.M) return;                              // (uniform; the grid is 8 * per_xcd)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fq = lane >> 4;
  const int ksteps = 9 * a.dg;
  const int offc = a.dg * 18;

  // ---- this thread's TWO pixels of the tile (fixed for the whole K loop) and its 16-byte chunk of their 64-channel rows: a wave's load
  // instruction then covers 8 pixels x 128 contiguous bytes (8 cache lines fully used; the (pixel, 32-byte quarter) mapping touched
  // 16 lines half-used per instruction and every line twice), and an LDS write covers 1 KiB contiguously like a DMA piece
  constexpr int HP = BM / 2;
  const int ck = tid & 7;
  int px2[2], h_in2[2], w_in2[2];
  long m2[2];
  bool live2[2];
  const bf16_t* xq2[2];
  const bf16_t* op2[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    px2[u] = (tid >> 3) + u * HP;
    m2[u] = tile * BM + px2[u];
    live2[u] = m2[u] < a.M;
    const long mm = live2[u] ? m2[u] : a.M - 1;
    const int wo = (int)(mm % a.Wo);
    const long mr = mm / a.Wo;
    const int ho = (int)(mr % a.Ho), n = (int)(mr / a.Ho);
    h_in2[u] = ho * a.stride - a.pad_t;
    w_in2[u] = wo * a.stride - a.pad_l;
    xq2[u] = a.x + ((long)n * a.H * a.W) * a.C + ck * 8;
    op2[u] = a.offs + mm * offc;
  }

  // ---- weight DMA: piece = 8 rows x 128 B; lane -> (row = piece*8 + lane/8, 16-byte chunk lane%8), chunk c of row r lands at c ^ (r & 7)
  const df_u32x4 rsrc_w = {(unsigned)(unsigned long long)a.w, (unsigned)((unsigned long long)a.w >> 32) & 0xFFFFu,
                           (unsigned)((a.Cout + 63) / 64 * 64) * (unsigned)a.Kpad * 2u, 0x00020000u};
  unsigned wvoff[WPW];
#pragma unroll
  for (int k = 0; k < WPW; ++k) {
    const int row = (wave * WPW + k) * 8 + (lane >> 3);
    wvoff[k] = (unsigned)row * (unsigned)a.Kpad * 2u + (unsigned)(((lane & 7) ^ (row & 7)) << 4);
  }
  auto issue_w = [&](int ks, int stage) __attribute__((always_inline)) {
    const unsigned lds = (unsigned)(2 * ABYTES + stage * WBYTES);
#pragma unroll
    for (int k = 0; k < WPW; ++k) df_dma16(rsrc_w, wvoff[k] + (unsigned)ks * 128u, lds + (unsigned)(wave * WPW + k) * 1024u);
  };

  // ---- gather of K-step ks: loads now, blend + LDS write later
  struct Taps { uint4 cn[2][4]; float wq[2][4]; bool in[2]; };         // per pixel: the four corner pieces (8 channels each), their weights
  // the offset pairs of a K-step are fetched one step before its gather (a dependent load in front of the eight corner loads would put a
  // memory latency at the head of every K-step)
  unsigned oraw_next[2];
  auto load_off = [&](int ks) __attribute__((always_inline)) {
    const int t = ks / a.dg, grp = ks - t * a.dg;
#pragma unroll
    for (int u = 0; u < 2; ++u) oraw_next[u] = *reinterpret_cast<const unsigned*>(op2[u] + (grp * 9 + t) * 2);
  };
  load_off(0);
  auto gather = [&](Taps& g, int ks) __attribute__((always_inline)) {
    const int t = ks / a.dg, grp = ks - t * a.dg;
    const int i = t / 3, j = t - i * 3;
    const unsigned oraw[2] = {oraw_next[0], oraw_next[1]};
    load_off(ks + 1 < ksteps ? ks + 1 : ks);                            // (always issued: the hand-counted waits below rely on 10 loads per gather)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int h_in = h_in2[u], w_in = w_in2[u];
      const int cur_h = a.H - h_in, cur_w = a.W - w_in;
      const float off_h = bf2f((bf16_t)(oraw[u] & 0xffffu)), off_w = bf2f((bf16_t)(oraw[u] >> 16));
      const float h_im = (float)(h_in + i * a.dil) + off_h;             // deform_conv.cu:261-262
      const float w_im = (float)(w_in + j * a.dil) + off_w;
      const bool in = h_im >= 0 && w_im >= 0 && h_im < a.H && w_im < a.W;         // :263
      g.in[u] = in;
      float mh = (float)(i * a.dil) + off_h, mw = (float)(j * a.dil) + off_w;     // :264-265 (relative to (h_in, w_in))
      if (!in) { mh = (float)(-h_in); mw = (float)(-w_in); }            // (any valid address: the result is discarded)
      int h_low = (int)floorf(mh), w_low = (int)floorf(mw), h_high, w_high;       // deformable_im2col_bilinear :94-112
      if (h_low >= cur_h - 1) { h_high = h_low = cur_h - 1; mh = (float)h_low; } else h_high = h_low + 1;
      if (w_low >= cur_w - 1) { w_high = w_low = cur_w - 1; mw = (float)w_low; } else w_high = w_low + 1;
      const float lh = mh - h_low, lw = mw - w_low, hh = 1 - lh, hw = 1 - lw;
      g.wq[u][0] = hh * hw; g.wq[u][1] = hh * lw; g.wq[u][2] = lh * hw; g.wq[u][3] = lh * lw;          // :118-125
      // (24-bit multiplies: rows, pixels of one image and C are all below 2^24 and an image below 2^32 elements — the launcher checks;
      // the 64-bit forms cost three quarter-rate v_mul_lo_u32 / v_mad_u64_u32 per corner in a loop that is VALU-bound)
      const bf16_t* base = xq2[u] + grp * 64;
      const unsigned rl = __umul24((unsigned)(h_in + h_low), (unsigned)a.W) + (unsigned)w_in;
      const unsigned rh = __umul24((unsigned)(h_in + h_high), (unsigned)a.W) + (unsigned)w_in;
      g.cn[u][0] = *reinterpret_cast<const uint4*>(base + __umul24(rl + (unsigned)w_low, (unsigned)a.C));
      g.cn[u][1] = *reinterpret_cast<const uint4*>(base + __umul24(rl + (unsigned)w_high, (unsigned)a.C));
      g.cn[u][2] = *reinterpret_cast<const uint4*>(base + __umul24(rh + (unsigned)w_low, (unsigned)a.C));
      g.cn[u][3] = *reinterpret_cast<const uint4*>(base + __umul24(rh + (unsigned)w_high, (unsigned)a.C));
    }
  };
  uint4 pend[2];
  int pend_ks = -1;
  auto blend_store = [&](const Taps& g, int ks, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float v1[8], v2[8], v3[8], v4[8], o[8];
      df_unpack8(g.cn[u][0], v1); df_unpack8(g.cn[u][1], v2); df_unpack8(g.cn[u][2], v3); df_unpack8(g.cn[u][3], v4);
#pragma unroll
      for (int e = 0; e < 8; ++e)       // (the explicit fma chain of deform_sample_fwd_kernel: identical column values)
        o[e] = fmaf(g.wq[u][3], v4[e], fmaf(g.wq[u][2], v3[e], fmaf(g.wq[u][1], v2[e], g.wq[u][0] * v1[e])));
      const bool keep = g.in[u] && live2[u];        // a select on the packed words (a branch around the blend costs more than the blend)
      uint4 outv;
      outv.x = keep ? pack2bf(o[0], o[1]) : 0u; outv.y = keep ? pack2bf(o[2], o[3]) : 0u;
      outv.z = keep ? pack2bf(o[4], o[5]) : 0u; outv.w = keep ? pack2bf(o[6], o[7]) : 0u;
      *reinterpret_cast<uint4*>(smem + stage * ABYTES + px2[u] * 128 + ((ck ^ (px2[u] & 7)) << 4)) = outv;
      pend[u] = outv;
    }
    pend_ks = ks;                                                       // the column-buffer copy is stored at the start of the next step
  };
  auto flush_col = [&]() __attribute__((always_inline)) {              // (there its stores are OLDER than the step's DMA in the wave's queue)
    if (a.col && pend_ks >= 0) {
      const int t = pend_ks / a.dg, grp = pend_ks - t * a.dg;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (live2[u]) *reinterpret_cast<uint4*>(a.col + (m2[u] * 9 + t) * a.C + grp * 64 + ck * 8) = pend[u];
    }
    pend_ks = -1;
  };

  // ---- fragment addresses: weights = A operand (row = output channel), pixels = B operand
  int waddr[NCT], xaddr[NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int row = wn * (BN / 4) + c * 16 + frow;
    waddr[c] = 2 * ABYTES + row * 128 + ((fq ^ (row & 7)) << 4);       // k-slice 1: ^ 64
  }
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int r = wm * 64 + p * 16 + frow;
    xaddr[p] = r * 128 + ((fq ^ (r & 7)) << 4);
  }
  f32x4 acc[NCT][NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int p = 0; p < NPT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: K-step 0 staged, the gather of K-step 1 in flight
  Taps s0, s1;                                      // K-step s blends from set s & 1
  gather(s0, 0);
  if (ksteps > 1) gather(s1, 1);
  issue_w(0, 0);
  blend_store(s0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // One K-step.  The corner rows are gathered TWO steps ahead (memory latency is several K-steps of MFMA time and all eight waves of the
  // single resident workgroup move in lockstep): queue of a wave at its wait, oldest first =
  //   [rows(ks+1) | column-buffer stores of the previous blend | W(ks+1) | rows(ks+2)]
  // so "W(ks+1) landed" = at most the 10 loads of rows(ks+2) outstanding.
  auto step = [&](int ks, const Taps& nxt, Taps& refill, int st) __attribute__((always_inline)) {
    const bool more = ks + 1 < ksteps, more2 = ks + 2 < ksteps;
    flush_col();
    if (more) issue_w(ks + 1, st ^ 1);
    if (more2) gather(refill, ks + 2);
    bf16x8 wf[2][NCT], xf[2][NPT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) wf[s2][c] = *reinterpret_cast<const bf16x8*>(smem + ((waddr[c] + st * WBYTES) ^ (s2 * 64)));
#pragma unroll
      for (int p = 0; p < NPT; ++p) xf[s2][p] = *reinterpret_cast<const bf16x8*>(smem + ((xaddr[p] + st * ABYTES) ^ (s2 * 64)));
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int p = 0; p < NPT; ++p) acc[c][p] = DH_MFMA_16x16x32(wf[s2][c], xf[s2][p], acc[c][p]);
    // (tried, round 5: the blend issued BETWEEN this step's MFMAs by sched_group_barrier — one MFMA, four VALU — instead of behind them:
    // 1.02 -> 1.16 ms without, 1.22 -> 1.45 ms with the column buffer; the second resident workgroup already fills the other pipe)
    __builtin_amdgcn_sched_barrier(0);
    if (more) blend_store(nxt, ks + 1, st ^ 1);
    if (!more2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  for (int ks = 0; ks < ksteps; ks += 2) {
    step(ks, s1, s0, 0);
    if (ks + 1 < ksteps) step(ks + 1, s0, s1, 1);
  }
  flush_col();

  // ---- epilogue: lane owns channels co0 + c*16 + fq*4 .. +3 of pixel (wm*64 + p*16 + frow)
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int co = wn * (BN / 4) + c * 16 + fq * 4;
    if (co >= a.Cout) continue;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
      const float4 t = *reinterpret_cast<const float4*>(a.bias + co);
      b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
    }
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const long mo = tile * BM + wm * 64 + p * 16 + frow;
      if (mo >= a.M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[c][p][r] + b4[r];
        if (a.relu) v[r] = fmaxf(v[r], 0.f);
      }
      uint2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(a.y + mo * a.Cout + co) = o;
    }
  }
}

template <int BN, int BM>
int launch_fused(const FusedArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * BM * 128 + 2 * BN * 128;
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&deform_fused_fwd_kernel<BN, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  const long tiles = (a.M + BM - 1) / BM;
  const long grid = (tiles + 7) / 8 * 8;                     // 8 XCDs x their share of the tiles (the kernel maps blockIdx -> tile)
  hipLaunchKernelGGL((deform_fused_fwd_kernel<BN, BM>), dim3((unsigned)grid), dim3(BM * 4), LDS, s, a);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

}  // namespace

bool danhip_deform_fused_eligible(int N, int H, int W, int C, int Cout, int kh, int kw, int stride, int dg) {
  if (!(kh == 3 && kw == 3 && dg > 0 && C % dg == 0 && C / dg == 64)) return false;
  if (Cout % 64 != 0 || Cout > 256 || Cout == 192) return false;         // one N tile = the whole Cout (64 / 128 / 256)
  const long Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  return (long)N * Ho * Wo * 9 * C < (1l << 40) && (long)N * H * W * C < (1l << 31) && (long)H * W < (1l << 24) && C < (1 << 24);
}

// Returns DANHIP_OK when launched.  `col` may be NULL (no column buffer is written).
int danhip_launch_deform_fused_fwd(const uint16_t* x, const uint16_t* wf_packed, int kpad, const float* bias, const uint16_t* offsets, uint16_t* y,
                                   uint16_t* col, int N, int H, int W, int C, int Cout, int stride, int dil, int dg, int relu, hipStream_t s) {
  FusedArgs a{};
  a.x = x; a.offs = offsets; a.w = wf_packed; a.bias = bias; a.y = y; a.col = col;
  a.N = N; a.H = H; a.W = W; a.C = C; a.Cout = Cout; a.Kpad = kpad; a.dg = dg; a.stride = stride; a.dil = dil; a.relu = relu;
  a.Ho = (H + stride - 1) / stride;
  a.Wo = (W + stride - 1) / stride;
  int th = (a.Ho - 1) * stride + 3 - H; if (th < 0) th = 0;            // SAME pad_before from the undilated kernel (deform_conv.cc:473-479)
  int tw = (a.Wo - 1) * stride + 3 - W; if (tw < 0) tw = 0;
  a.pad_t = th / 2; a.pad_l = tw / 2;
  a.M = (long)N * a.Ho * a.Wo;
  if (Cout == 256) return launch_fused<256, FUSED_BM>(a, s);
  if (Cout == 128) return launch_fused<128, FUSED_BM>(a, s);
  return launch_fused<64, FUSED_BM>(a, s);
}
