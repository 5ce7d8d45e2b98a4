// Weight gradient of POINTWISE (1x1 / stride 1) convolutions on MFMA for gfx950:
//
//   dW[ci][co] = sum over pixels  X[m][ci] * dY[m][co]          db[co] = sum over pixels dY[m][co]
//
// Reference call sites: every 1x1 of the LFPN / context modules / stage-2 mixing (net/pb_net.py:198-218, net/danet.py:352-372, 842-954,
// net/danet_deform.py:267-290) and the filter gradient of DeformConvBackpropOp over its sampled columns (cpp/Deform/deform_conv.cc:757-768:
// dW = dOut . col^T, here Cin = 9 C).  214 of DAN's convolutions are pointwise; their weight gradients ran on the generic per-tap
// split-K kernel (conv_wgrad.hip: 64 x 64 wave tiles re-read per tap, vmcnt(0) + barrier per tile) and were the largest line of the
// DAN-Deform step.
//
// Same streaming skeleton as the 3x3 row kernel (conv_wgrad_rows.hip): the reduction runs over pixels, both operands are staged exactly
// as they lie in HBM ([pixel][channel] rows, 16-byte LDS-DMA pieces, inline asm with counted vmcnt), fragments come from
// ds_read_b64_tr_b16, two wave groups alternate memory / MFMA phases.  A 512-thread workgroup owns a 256 ci x 256 co gradient tile
// (wave tile 128 ci x 64 co = 32 accumulators: 24 transposing reads per 32 MFMAs) and walks its share of 32-pixel K-steps through a
// 4-deep LDS ring (X: four [32 px][64 ci] sub-tiles, dY: two [32 px][128 co] sub-tiles per step, the swizzles of the row kernel).
// Waves whose ci / co range lies beyond the layer's channels skip their MFMAs; partial sums are combined with fp32 atomics.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace {

struct WgPwArgs {
  const bf16_t* x;     // [M,C]
  const bf16_t* dy;    // [M,Co8]
  float* dw;           // [cin_real,Cout]
  float* db;           // [Cout] or null
  int M, C, Co8, Cout, cin_real;
  int ldx, ldy;        // pixel pitches of x / dy in elements (channel-slice views; dense: C / Co8)
  int ksteps, steps_per_split, ci_tiles, co_tiles, xcd_grouped;
  float* slab;         // optional: partial tiles as plain stores + wg_pw_reduce_kernel (short launches; see conv_wgrad_rows.hip)
  int splits;
  int b2;              // 1: second barrier per K-step (option "wgrad_b2")
  FastDiv div_ci, div_pairs;
};

typedef __attribute__((ext_vector_type(4))) unsigned wp_u32x4;
template <int N>
__device__ __forceinline__ void wp_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wp_dma16(wp_u32x4 rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ wp_u32x4 wp_make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return wp_u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ int wp_f128(int pc) { return ((pc >> 1) & 1) | (((pc >> 3) & 1) << 1); }
__device__ __forceinline__ int wp_f256(int px) { return (px & 3) | (((px >> 3) & 1) << 2); }

__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_pw_kernel(const WgPwArgs a) {
  constexpr int XSUB = 32 * 128;                      // one [32 px][64 ci] sub-tile
  constexpr int YSUB = 32 * 256;                      // one [32 px][128 co] sub-tile
  constexpr int XS = 4 * XSUB, YS = 2 * YSUB;         // per K-step: 16 KB + 16 KB
  constexpr int DEPTH = 4, P = 2;                     // ring depth = unroll; the DMA runs P K-steps ahead
  constexpr int YBASE = 0, XBASE = DEPTH * YS;
  constexpr int DPW = 4;                              // DMA instructions per wave and K-step (2 X pieces + 2 dY pieces)
  static_assert(XBASE + DEPTH * XS <= 160 * 1024, "LDS budget");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                          // phase group: 0 = A, 1 = B
  const int wci = grp;                                // ci half of the block tile (128 ci): the two phase groups
  const int wco = wave & 3;                           // co quarter (64 co)

  const int pairs = a.ci_tiles * a.co_tiles;
  int split, pair;
  if (a.xcd_grouped) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int sq = (int)fdiv((unsigned)slot, a.div_pairs);
    pair = slot - sq * pairs;
    split = sq * 8 + xcd;
  } else {
    split = (int)fdiv(blockIdx.x, a.div_pairs);
    pair = (int)blockIdx.x - split * pairs;
  }
  const int co_tile = (int)fdiv((unsigned)pair, a.div_ci);
  const int ci_tile = pair - co_tile * a.ci_tiles;
  const int ci0 = ci_tile * 256, co0 = co_tile * 256;
  const int k_begin = split * a.steps_per_split;
  const int k_end = min(a.ksteps, k_begin + a.steps_per_split);
  if (k_begin >= k_end) return;
  const int V = (k_end - k_begin + DEPTH - 1) / DEPTH * DEPTH;

  const wp_u32x4 rsrc_x = wp_make_rsrc(a.x, ((unsigned)(a.M - 1) * (unsigned)a.ldx + (unsigned)a.C) * 2u);          // pixels >= M: out of range, zero fill
  const wp_u32x4 rsrc_y = wp_make_rsrc(a.dy, ((unsigned)(a.M - 1) * (unsigned)a.ldy + (unsigned)a.Co8) * 2u);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS char*)smem);

  // ---- DMA lane constants: wave w moves X pieces 2w, 2w+1 (sub-tile p >> 2, pixels 8 (p & 3) ..) and dY pieces 2w, 2w+1 (sub-tile
  // p >> 3, pixels 4 (p & 7) ..) of every K-step
  unsigned vx[2], vy[2];
  int xdst[2], ydst[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    {
      const int pz = wave * 2 + k, sub = pz >> 2, pp = pz & 3;
      const int l8 = lane >> 3, pcol = pp * 8 + l8;
      const int chunk = (lane & 7) ^ (wp_f128(pcol) << 1);
      const int ci = ci0 + sub * 64;
      vx[k] = ci < a.C ? (unsigned)pcol * (unsigned)(a.ldx * 2) + (unsigned)(ci * 2 + (chunk << 4)) : 0xFFFFFFFFu;
      xdst[k] = XBASE + sub * XSUB + pp * 1024;
    }
    {
      const int pz = wave * 2 + k, sub = pz >> 3, pp = pz & 7;
      const int lp = lane >> 4, cpos = lane & 15, px = pp * 4 + lp;
      const int chunk = cpos ^ (wp_f256(px) << 1);
      const int co = co0 + sub * 128 + chunk * 8;
      vy[k] = co < a.Co8 ? (unsigned)px * (unsigned)(a.ldy * 2) + (unsigned)(co * 2) : 0xFFFFFFFFu;
      ydst[k] = YBASE + sub * YSUB + pp * 1024;
    }
  }
  auto dma_step = [&](auto slc, int kstep) __attribute__((always_inline)) {      // K-step kstep (32 pixels) into ring slot SL
    constexpr int SL = decltype(slc)::value;
    const unsigned inv = kstep < k_end ? 0u : 0xFFFFFFFFu;                       // beyond this block's share: zero fill
    const unsigned sx = (unsigned)(kstep * 32) * (unsigned)(a.ldx * 2), sy = (unsigned)(kstep * 32) * (unsigned)(a.ldy * 2);
#pragma unroll
    for (int k = 0; k < 2; ++k) wp_dma16(rsrc_x, (vx[k] + sx) | (vx[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | inv, lds0 + SL * XS + xdst[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) wp_dma16(rsrc_y, (vy[k] + sy) | (vy[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | inv, lds0 + SL * YS + ydst[k]);
  };

  // ---- fragment addresses of ring slot 0
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  // X fragment i (ci 16 i .. of this wave's half), pixel half h:
  //   XBASE + (wci*2 + (i>>2)) * XSUB + pcol*128 + ((((i&3)*2 + (p>>1)) ^ (f128(pcol) << 1)) << 4) + (p&1)*8,   pcol = 8g + 4h + q
  int xbase[2], xkey[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pcol = 8 * g + 4 * h + q;
    xbase[h] = XBASE + pcol * 128 + (p & 1) * 8;
    xkey[h] = wp_f128(pcol) << 1;                     // swizzle key of this lane's pixel row (chunk units)
  }
  int yaddr[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int px = 8 * g + q;
    const int ch = (((wco & 1) * 4 + o) * 2 + (p >> 1)) ^ (wp_f256(px) << 1);
    yaddr[o] = YBASE + (wco >> 1) * YSUB + px * 256 + (ch << 4) + (p & 1) * 8;
  }
  int xoff[8][2];                                     // [ci fragment i][pixel half]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      xoff[i][h] = xbase[h] + (wci * 2 + (i >> 2)) * XSUB + ((((i & 3) * 2 + (p >> 1)) ^ xkey[h]) << 4);

  const bool active = ci0 + wci * 128 < a.C && co0 + wco * 64 < a.Co8;      // wave-uniform: this wave's 128 ci x 64 co hold real channels

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[i][o] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.db != nullptr && ci_tile == 0 && co0 + wco * 64 < a.Co8;       // wave (wci, wco) sums co fragments 2 wci, 2 wci + 1 of its quarter
  f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  bf16x8 xf[8], yf[4];
  auto tr2 = [&](int addr) __attribute__((always_inline)) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(smem + addr));
  };
  auto mem = [&](auto uc) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const s16x4 lo = tr2(yaddr[o] + U * YS), hi = tr2(yaddr[o] + 4 * 256 + U * YS);
      yf[o] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const s16x4 lo = tr2(xoff[i][0] + U * XS), hi = tr2(xoff[i][1] + U * XS);
      xf[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  auto mma = [&]() __attribute__((always_inline)) {
    // (fragments beyond the layer's channels hold zeros — the DMA zero-fills them — so a ragged tile is computed whole; a wave with
    // no real channels at all skips the phase)
    if (active) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[i][o] = DH_MFMA_16x16x32(xf[i], yf[o], acc[i][o]);
    }
    if (do_bias) {
      const act16_t one = (act16_t)1.0f;
      const bf16x8 ones = {one, one, one, one, one, one, one, one};
      if (wci == 0) { accb[0] = DH_MFMA_16x16x32(ones, yf[0], accb[0]); accb[1] = DH_MFMA_16x16x32(ones, yf[1], accb[1]); }
      else          { accb[0] = DH_MFMA_16x16x32(ones, yf[2], accb[0]); accb[1] = DH_MFMA_16x16x32(ones, yf[3], accb[1]); }
    }
  };

  // ---- prologue: K-steps 0 .. P-1
  dma_step(std::integral_constant<int, 0>{}, k_begin);
  dma_step(std::integral_constant<int, 1>{}, k_begin + 1);
  static_assert(P == 2, "prologue issues P steps");
  wp_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // K-step v (slot U = v % 4):   A:  DMA(v+P) + mem(v) | b1 | MFMA(v) | b2        B:  MFMA(v) | b1 | DMA(v+1+P) + mem(v+1) | b2
  // A wave's pieces of K-step u are waited for (vmcnt((P-1) DPW)) before b1 of cycle u-1; B reads them in the phase after that barrier.
  // Slot (v+P) % 4 was last read two cycles earlier.
  if (grp == 0) {
    for (int v = 0; v < V; v += DEPTH) {
      auto step = [&](auto uc) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        dma_step(std::integral_constant<int, (U + P) % DEPTH>{}, k_begin + v + U + P);
        __builtin_amdgcn_sched_barrier(0);
        mem(uc);
        __builtin_amdgcn_sched_barrier(0);
        wp_wait_vmcnt<(P - 1) * DPW>();
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();                // b1
        __builtin_amdgcn_sched_barrier(0);
        if (k_begin + v + U < k_end) mma();
        __builtin_amdgcn_sched_barrier(0);
        if (a.b2) __builtin_amdgcn_s_barrier();      // b2
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    }
  } else {
    mem(std::integral_constant<int, 0>{});
    dma_step(std::integral_constant<int, P % DEPTH>{}, k_begin + P);
    for (int v = 0; v < V; v += DEPTH) {
      auto step = [&](auto uc) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        if (k_begin + v + U < k_end) mma();
        __builtin_amdgcn_sched_barrier(0);
        wp_wait_vmcnt<(P - 1) * DPW>();
        __builtin_amdgcn_s_barrier();                // b1
        dma_step(std::integral_constant<int, (U + 1 + P) % DEPTH>{}, k_begin + v + U + 1 + P);
        __builtin_amdgcn_sched_barrier(0);
        mem(std::integral_constant<int, (U + 1) % DEPTH>{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (a.b2) __builtin_amdgcn_s_barrier();      // b2
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    }
  }
  wp_wait_vmcnt<0>();

  // ---- epilogue: lane holds dW[ci = ci0 + wci*128 + i*16 + g*4 + r][co = co0 + wco*64 + o*16 + (lane & 15)]
  if (a.slab) {                                        // slab form: register order, 1 KiB per wave-instruction, slots [i*4 + o][wave][lane]
    if (active) {
      f32x4* dst = reinterpret_cast<f32x4*>(a.slab) + (size_t)blockIdx.x * (32 * 512) + wave * 64 + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int o = 0; o < 4; ++o) __builtin_nontemporal_store(acc[i][o], dst + (i * 4 + o) * 512);
    }
  } else if (active)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + wci * 128 + i * 16 + g * 4 + r;
      if (ci >= a.cin_real) continue;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int co = co0 + wco * 64 + o * 16 + (lane & 15);
        if (co < a.Cout) atomicAdd(a.dw + ((size_t)ci * a.Cout + co), acc[i][o][r]);
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int co = co0 + wco * 64 + (wci * 2 + e) * 16 + (lane & 15);
      if (lane < 16 && co < a.Cout) atomicAdd(a.db + co, accb[e][0]);
    }
  }
}

// Second pass of the slab form (the combine of conv_wgrad_rows.hip for the 256 x 256 tile): a workgroup owns 64 consecutive float4 slots,
// wave w sums the splits w, w + 4, ..., wave 0 adds the total into dW[ci][co].  Slots of waves without real channels were never written
// and are skipped (the same test as the kernel's `active`).
__global__ __launch_bounds__(256) void wg_pw_reduce_kernel(const WgPwArgs a) {
  constexpr int TILE = 32 * 512;
  __shared__ f32x4 part[4][64];
  const int pairs = a.ci_tiles * a.co_tiles;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int chunk = blockIdx.x;
  const int pair = chunk / (TILE / 64), q = (chunk - pair * (TILE / 64)) * 64 + lane;
  const int io = q >> 9, wave = (q >> 6) & 7;
  const int wci = wave >> 2, wco = wave & 3;
  const int co_tile = pair / a.ci_tiles, ci_tile = pair - co_tile * a.ci_tiles;
  const int ci0 = ci_tile * 256, co0 = co_tile * 256;
  if (!(ci0 + wci * 128 < a.C && co0 + wco * 64 < a.Co8)) return;       // (uniform over the workgroup: `wave` is the same for its 64 slots)
  const f32x4* src = reinterpret_cast<const f32x4*>(a.slab) + q;
  auto blk = [&](int split) -> size_t {
    return a.xcd_grouped ? (size_t)(((split >> 3) * pairs + pair) * 8 + (split & 7)) : (size_t)(split * pairs + pair);
  };
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int sp = w;
  for (; sp + 28 < a.splits; sp += 32) {
    f32x4 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_nontemporal_load(src + blk(sp + 4 * e) * TILE);
    s0 += (v[0] + v[1]) + (v[2] + v[3]);
    s1 += (v[4] + v[5]) + (v[6] + v[7]);
  }
  for (; sp < a.splits; sp += 4) s0 += __builtin_nontemporal_load(src + blk(sp) * TILE);
  part[w][lane] = s0 + s1;
  __syncthreads();
  if (w != 0) return;
  const f32x4 sum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  const int i = io >> 2, o = io & 3, g = lane >> 4;
  const int co = co0 + wco * 64 + o * 16 + (lane & 15);
  if (co >= a.Cout) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ci = ci0 + wci * 128 + i * 16 + g * 4 + r;
    if (ci < a.cin_real) a.dw[(size_t)ci * a.Cout + co] += sum[r];
  }
}

int wp_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

}  // namespace

static bool wg_pw_eligible(const danhip_conv_desc* d) {
  if (!(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W)) return false;
  const int co8 = (d->Cout + 7) / 8 * 8;
  if (d->Cin % 64 != 0 || d->Cin < 128 || co8 < 64) return false;          // thin operands: the generic kernel's 64-wide tiles waste less
  const long M = (long)d->N * d->H * d->W;
  if (M < 4096) return false;
  if (M * d->Cin >= (1l << 31) || M * co8 >= (1l << 31)) return false;
  return true;
}

const char* danhip_wgrad_pw_label(const danhip_conv_desc* d) { return wg_pw_eligible(d) ? "conv_wgrad_pw_kernel" : nullptr; }

size_t danhip_wgrad_pw_workspace_bytes(const danhip_conv_desc* d) {
  if (!wg_pw_eligible(d)) return 0;
  const int co8 = (d->Cout + 7) / 8 * 8;
  const int pairs = ((d->Cin + 255) / 256) * ((co8 + 255) / 256);
  const int ksteps = (d->N * d->H * d->W + 31) / 32;
  int splits = wp_cu_count() / pairs;
  if (splits < 1) splits = 1;
  if (splits > ksteps) splits = ksteps;
  const int steps_per_split = (ksteps + splits - 1) / splits;
  const int slab_mode = danhip_option("wgrad_slab");
  if (splits < 2 || (slab_mode != 2 && steps_per_split > 192)) return 0;      // long launches keep the atomic epilogue
  return (size_t)wp_cu_count() * 32 * 512 * 16;
}

// Returns DANHIP_OK when launched, 1 when the shape is not eligible (caller falls back to conv_wgrad.hip).
int danhip_launch_wgrad_pw(const danhip_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* db, int cin_real, hipStream_t s,
                           void* ws, size_t ws_bytes, int ldx, int ldy) {
  if (!wg_pw_eligible(d)) return 1;
  {   // a channel's 16-byte chunk offset inside a pixel row must keep the row's swizzle: within the 64-channel (X) / 128-channel (dY) sub-tile
    const long M_ = (long)d->N * d->H * d->W;
    if (ldx && M_ * ldx >= (1l << 31)) return 1;
    if (ldy && M_ * ldy >= (1l << 31)) return 1;
  }
  constexpr int LDS = 4 * (16384 + 16384);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_pw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  WgPwArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.db = db;
  a.M = d->N * d->H * d->W; a.C = d->Cin; a.Co8 = (d->Cout + 7) / 8 * 8; a.Cout = d->Cout; a.cin_real = cin_real;
  a.ldx = ldx ? ldx : a.C; a.ldy = ldy ? ldy : a.Co8;
  a.ksteps = (a.M + 31) / 32;
  a.ci_tiles = (a.C + 255) / 256;
  a.co_tiles = (a.Co8 + 255) / 256;
  const int pairs = a.ci_tiles * a.co_tiles;
  int splits = wp_cu_count() / pairs;
  if (splits < 1) splits = 1;
  if (splits > a.ksteps) splits = a.ksteps;
  a.steps_per_split = (a.ksteps + splits - 1) / splits;
  splits = (a.ksteps + a.steps_per_split - 1) / a.steps_per_split;
  a.div_ci = make_fastdiv(a.ci_tiles);
  a.div_pairs = make_fastdiv(pairs);
  a.xcd_grouped = (splits % 8 == 0 && pairs > 1) ? 1 : 0;
  a.splits = splits;
  a.b2 = danhip_option("wgrad_b2");
  // slab form for short launches only (see conv_wgrad_rows.hip: on long ones the atomic tail hides under the other blocks' MFMAs)
  const int slab_mode = danhip_option("wgrad_slab");
  a.slab = (slab_mode && ws && ws_bytes >= danhip_wgrad_pw_workspace_bytes(d) && splits >= 2 && (slab_mode == 2 || a.steps_per_split <= 192))
               ? reinterpret_cast<float*>(ws) : nullptr;
  hipLaunchKernelGGL(conv_wgrad_pw_kernel, dim3(pairs * splits), dim3(512), LDS, s, a);
  DH_LAUNCH_CHECK();
  if (a.slab) {
    hipLaunchKernelGGL(wg_pw_reduce_kernel, dim3(pairs * (32 * 512 / 64)), dim3(256), 0, s, a);
    DH_LAUNCH_CHECK();
  }
  return DANHIP_OK;
}
