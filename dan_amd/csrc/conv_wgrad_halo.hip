// Weight gradient of 3x3 / stride-1 convolutions on MFMA for gfx950 — halo-reuse, register-resident 9-tap accumulation.
//
//   dW[tap][ci][co] = sum over pixels  X[pixel + tap][ci] * dY[pixel][co]         (tf.layers.conv2d backward w.r.t. the
//   kernel variable of net/sfd_net.py:81-89 and every 3x3 conv of net/*.py; bias gradient db[co] = sum dY rides along)
//
// The reduction runs over PIXELS, the strided axis of both NHWC operands, so both tiles are staged exactly as they lie in
// HBM ([pixel][channel] rows, 16-byte LDS-DMA pieces) and MFMA fragments come from ds_read_b64_tr_b16 (a 4-pixel x
// 16-channel block per 16-lane group): nothing is transposed in registers or through HBM.
//
// A persistent 512-thread workgroup owns a (64 ci) x (COT co) x (9 taps) gradient tile IN REGISTERS (wave tile 16 ci x
// 64 co x 9 taps = 144 accumulator VGPRs) and sweeps its share of 4 x 32 pixel tiles.  Per pixel tile the X halo patch
// (6 x 34 pixels x 64 ci, LDS row pitch 40) and the dY tile (128 pixels x COT co) are DMA'd once; every K-step (one 32-pixel row) reads the dY
// fragments once and reuses them for all nine taps, whose X fragments are the same patch at shifted row/column offsets
// (the LDS swizzle is keyed on the patch COLUMN so a tap's row shift is a pure immediate offset).  Compared with the
// per-tap split-K kernel (conv_wgrad.hip) this moves ~5.6x fewer bytes L2->LDS per MAC and reads 0.7 fragments per MFMA.
//
// Two wave groups alternate phases exactly as in conv_halo.hip: while one group issues its 36 MFMAs with nothing else in
// its stream, the other does its LDS reads / DMA issue; two barriers per cycle swap the roles.
// Partial gradients are combined with fp32 atomics into the HWIO tensor (zeroed by the caller once per step).
#include <type_traits>

#include "conv_common.h"

namespace {

struct WgHaloArgs {
  const bf16_t* x;     // [N,H,W,C]
  const bf16_t* dy;    // [N,H,W,Co8]
  float* dw;           // [3,3,cin_real,Cout]
  float* db;           // [Cout] or null
  int N, H, W, C, Co8, Cout, cin_real;
  int tiles_x, tiles_y, total_tiles, tiles_per_split;
  int ci_tiles, co_tiles, xcd_grouped;
  FastDiv div_tx, div_txy, div_ci, div_pairs;
};

template <int N>
__device__ __forceinline__ void wg_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ int f128(int pc) { return ((pc >> 1) & 1) | (((pc >> 3) & 1) << 1); }      // 32-byte-slot swizzles
__device__ __forceinline__ int f256(int px) { return (px & 3) | (((px >> 3) & 1) << 2); }

// COT: output-channel tile (128: waves = 4 ci-groups x 2 co-halves, every wave sees every K-step;
//                           64: waves = 4 ci-groups x 2 K-groups, group A takes the even rows, group B the odd ones)
#ifndef WG_ALL9
#define WG_ALL9 1
#endif
template <int COT, bool FRONT = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_halo_kernel(const WgHaloArgs a) {
  constexpr int TH = 4, TW = 32, PW = TW + 2;
  // The X halo patch ((TH+2) x PW pixels, 128-byte rows = 64 ci) lives in LDS with a row pitch of PITCH = 40 pixels = 5 DMA
  // pieces: a piece (8 pixels) never straddles two patch rows, so its row / validity / base address are wave-uniform
  // (scalar ALU) and a lane adds only its pixel-in-piece and chunk — ~7 VALU per DMA instead of ~27 (divisions by 34 and
  // three quarter-rate 32-bit multiplies per lane), which was 14 % of the kernel's time.  Pieces 4 of each row carry 2 pixels.
  constexpr int PITCH = 40, PPR = PITCH / 8;          // pieces per patch row
  constexpr int XPIECES = (TH + 2) * PPR;             // 30
  constexpr int XBYTES = XPIECES * 1024;
  constexpr int RBY = COT * 2;                        // dY row bytes
  constexpr int YBYTES = TH * TW * RBY;
  constexpr int YPIECES = YBYTES / 1024;              // 32 / 16
  constexpr int YROWS_PER_PIECE = 1024 / RBY;         // 4 / 8
  constexpr int BUF = XBYTES + YBYTES;
  constexpr int KPC = COT == 128 ? 1 : 2;             // K-steps (pixel rows) per cycle: one per group when the groups split K
  constexpr int CPT = TH / KPC;                       // cycles per pixel tile
  constexpr int NX = (XPIECES + 7) / 8, NY = YPIECES / 8;
  constexpr int NDMA = NX + NY;                       // DMA instructions per wave per tile
  constexpr int PER_EVEN = (NDMA + (CPT - 1) - 1) / (CPT - 1);     // spread over cycles 0 .. CPT-2 ...
  constexpr int PER_SPREAD = (NX == NY && 2 <= CPT - 1) ? NX : PER_EVEN;   // ... as one X-only and one dY-only call when that fits
  constexpr int PER = FRONT ? NDMA : PER_SPREAD;                   // FRONT: everything in cycle 0 (measured 1-4 % slower: kept off)
  constexpr int NO = 4;                               // co fragments per wave (64 co)
  static_assert(2 * BUF <= 160 * 1024, "LDS budget");
  static_assert(5 * PITCH * 128 + BUF < 65536 + BUF, "imm offsets");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int lane_ = lane;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave & 3;
  const int grp = wave >> 2;                          // phase group: 0 = A, 1 = B
  const int wco = COT == 128 ? grp : 0;

  // ---- block -> (ci tile, co tile, pixel split)
  // The (ci, co) tile pairs of one pixel split read the same X / dY tiles: put them on one XCD (blocks b and b+8 share an
  // XCD under round-robin dispatch; speed only) when the grid divides evenly, so the second..last readers hit that L2.
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
  const int ci0 = ci_tile * 64, co0 = co_tile * COT;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.total_tiles, t_begin + a.tiles_per_split);
  if (t_begin >= t_end) return;

  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)a.C * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dy), 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)a.Co8 * 2u), 0x00020000);

  // ---- DMA geometry: slot s < NX is X piece (s*8 + wave, clamped), else dY piece ((s-NX)*8 + wave)
  auto tile_coords = [&](int t, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    n = (int)fdiv((unsigned)t, a.div_txy);
    const int rem = t - n * (a.tiles_x * a.tiles_y);
    const int ty = (int)fdiv((unsigned)rem, a.div_tx);
    y0 = ty * TH;
    x0 = (rem - ty * a.tiles_x) * TW;
  };
  int dn = 0, dy0 = 0, dx0 = 0;                        // coordinates of the tile whose pieces are being issued (uniform)
  auto dma_setup = [&](int t) __attribute__((always_inline)) {
    tile_coords(t, dn, dy0, dx0);
    dn = __builtin_amdgcn_readfirstlane(dn); dy0 = __builtin_amdgcn_readfirstlane(dy0); dx0 = __builtin_amdgcn_readfirstlane(dx0);
  };
  auto dma_issue = [&](int bufoff, auto s0c, auto s1c) __attribute__((always_inline)) {      // slots [S0, S1)
    int lane = lane_;                                 // opaque copy: the lane-only terms below are recomputed per call (~8 VALU)
    asm volatile("" : "+v"(lane));                    // instead of being hoisted out of the tile loop into long-lived VGPRs
    int wv = wave;                                    // same for the wave-uniform per-slot geometry (would pin ~40 SGPRs and
    asm volatile("" : "+s"(wv));                      // spill them into VGPR lanes)
    constexpr bool HAS_X = decltype(s0c)::value < NX, HAS_Y = decltype(s1c)::value > NX;
    constexpr int CPP = RBY / 16;                                        // dY: 16-byte chunks per pixel (16 / 8)
    int l8 = 0, xA = 0, lp = 0, yB = 0;
    if constexpr (HAS_X) {
      l8 = lane >> 3;                                                    // X: pixel inside the piece
      xA = (lane & 7) ^ (((l8 >> 1) & 1) << 1);                          // X: chunk position, lane part of the swizzle key
    }
    if constexpr (HAS_Y) {
      lp = lane / CPP;                                                   // dY: pixel inside the piece
      const int cpos = lane % CPP;                                       // dY: chunk position
      yB = COT == 128 ? (cpos ^ (lp << 1)) : (cpos ^ (((lp >> 1) & 1) << 1));
    }
    const bool need_cc = co0 + COT > a.Co8;                              // thin heads: channel chunks beyond Co8 are zero-filled
#pragma unroll
    for (int s = decltype(s0c)::value; s < decltype(s1c)::value && s < NDMA; ++s) {
      if (s < NX) {
        int piece = s * 8 + wv;                      // wave-uniform from here to sbase
        if (piece > XPIECES - 1) piece = XPIECES - 1;
        const int prow = piece / PPR, pc5 = piece - prow * PPR;
        const int y = dy0 - 1 + prow, xs = dx0 - 1 + pc5 * 8;
        const unsigned bady = (unsigned)y < (unsigned)a.H ? 0u : 0xFFFFFFFFu;                  // scalar select, no branch
        const int lim = pc5 == PPR - 1 ? PW - (PPR - 1) * 8 : 8;
        const unsigned sbase = (unsigned)(((dn * a.H + y) * a.W + xs) * a.C + ci0) * 2u;      // may wrap; exact for valid lanes
        const int x = xs + l8;
        const unsigned badx = (l8 < lim && (unsigned)x < (unsigned)a.W) ? 0u : 0xFFFFFFFFu;
        const int chunk = xA ^ ((pc5 & 1) << 2);       // = (lane & 7) ^ (f128(pcol) << 1), pcol = pc5*8 + l8
        const unsigned voff = (__umul24((unsigned)l8, (unsigned)(a.C * 2)) + sbase + (unsigned)(chunk << 4)) | bady | badx;   // invalid -> ~0
        wg_dma16(rsrc_x, voff, smem + bufoff + piece * 1024);
        __builtin_amdgcn_sched_barrier(0);             // one slot's temporaries at a time (register budget)
      } else {
        const int piece = (s - NX) * 8 + wv;         // wave-uniform
        const int pxb = piece * YROWS_PER_PIECE;       // tile pixel r*32 + c of the piece's first pixel
        const int y = dy0 + (pxb >> 5), xs = dx0 + (pxb & 31);
        const unsigned bady = y < a.H ? 0u : 0xFFFFFFFFu;
        const unsigned sbase = (unsigned)(((dn * a.H + y) * a.W + xs) * a.Co8 + co0) * 2u;
        const int x = xs + lp;
        const int sb = (pxb >> 3) & 1;
        const int chunk = yB ^ (COT == 128 ? (sb << 3) : (sb << 2));     // = cpos ^ (f256 / f128 (pxb + lp) << 1)
        const int cclim = need_cc ? a.Co8 : 0x7FFFFFFF;                  // thin heads: chunks beyond Co8 are zero-filled
        const unsigned badx = (x < a.W && co0 + chunk * 8 < cclim) ? 0u : 0xFFFFFFFFu;
        const unsigned voff = (__umul24((unsigned)lp, (unsigned)(a.Co8 * 2)) + sbase + (unsigned)(chunk << 4)) | bady | badx;
        wg_dma16(rsrc_y, voff, smem + bufoff + XBYTES + piece * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- fragment addresses (buffer 0; flipped in place per tile)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  int xaddr[3][2];                                    // [tap column j][half h]: X fragment base, row offset is an immediate
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pcol = j + 8 * g + 4 * h + q;
      const int ch = (wci * 2 + (p >> 1)) ^ (f128(pcol) << 1);
      xaddr[j][h] = pcol * 128 + (ch << 4) + (p & 1) * 8;
    }
  int yaddr[NO];                                      // [co fragment]: dY fragment base of row 0, half 0; half 1 (pixel + 4) is
#pragma unroll                                         // + 4*RBY: the swizzle bits (px & 3, px >> 3) do not see bit 2 of the pixel
  for (int o = 0; o < NO; ++o) {
    const int px = 8 * g + q;                          // + r*32 (does not change the swizzle bits 0,1,3)
    const int ch = ((wco * 4 + o) * 2 + (p >> 1)) ^ ((COT == 128 ? f256(px) : f128(px)) << 1);
    yaddr[o] = XBYTES + px * RBY + (ch << 4) + (p & 1) * 8;
  }

  f32x4 acc[9][NO];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[t][o] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.db != nullptr && ci_tile == 0 && wci == 0;     // wave-uniform
  float dbsum[NO] = {0.f, 0.f, 0.f, 0.f};

  // Register budget: the X fragments of taps 0..5 are read in the mem phase; taps 6..8 are read at the start of the MFMA
  // phase into the registers of taps 0..2 once their MFMAs have been issued (their LDS latency hides under taps 3..5).
  constexpr int NXF = WG_ALL9 ? 9 : 6;
  bf16x8 xf[NXF], yf[NO];
  auto read_x = [&](bf16x8& dst, int tap, int R) __attribute__((always_inline)) {
    const int i = tap / 3, j = tap % 3;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(smem + xaddr[j][0] + (R + i) * PITCH * 128));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(smem + xaddr[j][1] + (R + i) * PITCH * 128));
    dst = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto read_frags = [&](auto rc) __attribute__((always_inline)) {        // K-step R of the current tile (addresses hold the buffer)
    constexpr int R = decltype(rc)::value;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(smem + yaddr[o] + R * 32 * RBY));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(smem + yaddr[o] + 4 * RBY + R * 32 * RBY));
      yf[o] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int t = 0; t < NXF; ++t) read_x(xf[t], t, R);
  };
  auto mma_tap = [&](int t, const bf16x8& x) __attribute__((always_inline)) {
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[t][o] = DH_MFMA_16x16x32(x, yf[o], acc[t][o]);
  };
  auto mma = [&](auto rc) __attribute__((always_inline)) {              // MFMA phase of K-step R
    constexpr int R = decltype(rc)::value;
    if constexpr (WG_ALL9) {
#pragma unroll
      for (int t = 0; t < 9; ++t) mma_tap(t, xf[t]);
    } else {
      mma_tap(0, xf[0]); mma_tap(1, xf[1]); mma_tap(2, xf[2]);
      __builtin_amdgcn_sched_barrier(0);
      read_x(xf[0], 6, R); read_x(xf[1], 7, R); read_x(xf[2], 8, R);
      __builtin_amdgcn_sched_barrier(0);
      mma_tap(3, xf[3]); mma_tap(4, xf[4]); mma_tap(5, xf[5]);
      __builtin_amdgcn_sched_barrier(0);
      mma_tap(6, xf[0]); mma_tap(7, xf[1]); mma_tap(8, xf[2]);
    }
    if (do_bias) {                                    // db += column sums of this K-step's dY fragments (VALU, beside the MFMAs)
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)yf[o][e];
        dbsum[o] += s;
      }
    }
  };
  auto flip_buffers = [&](int dir) __attribute__((always_inline)) {     // dir = +BUF or -BUF
#pragma unroll
    for (int j = 0; j < 3; ++j) { xaddr[j][0] += dir; xaddr[j][1] += dir; }
#pragma unroll
    for (int o = 0; o < NO; ++o) yaddr[o] += dir;
  };

  // ---- prologue: tile t_begin into buffer 0
  dma_setup(t_begin);
  dma_issue(0, std::integral_constant<int, 0>{}, std::integral_constant<int, NDMA>{});
  wg_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // Cycle c of a tile: group A works on K-step c*KPC, group B on K-step c*KPC + KPC-1.
  //   A:  mem(c) | b1 | MFMA(c) | b2          B:  MFMA(c) | b1 | mem(c+1) | b2
  // Next tile's DMA (other buffer) is issued in cycles 0..CPT-2 and retired (vmcnt(0)) before b1 of the last cycle, after
  // which B reads the next tile's first fragments.
  if (grp == 0) {
    for (int t = t_begin; t < t_end; ++t) {
      const int cur = (t - t_begin) & 1;
      const bool has_next = t + 1 < t_end;
      auto cycle = [&](auto cc) __attribute__((always_inline)) {
        constexpr int CYC = decltype(cc)::value;
        read_frags(std::integral_constant<int, CYC * KPC>{});
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
          if (CYC == 0) dma_setup(t + 1);
          if (CYC < CPT - 1) dma_issue((cur ^ 1) * BUF, std::integral_constant<int, CYC * PER>{}, std::integral_constant<int, (CYC + 1) * PER>{});
        }
        if (CYC == CPT - 1) wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();                // b1
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, CYC * KPC>{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                // b2
      };
      cycle(I0{});
      if constexpr (CPT > 1) cycle(I1{});
      if constexpr (CPT > 2) { cycle(I2{}); cycle(I3{}); }
      flip_buffers(cur == 0 ? BUF : -BUF);
    }
  } else {
    read_frags(std::integral_constant<int, KPC - 1>{});
    for (int t = t_begin; t < t_end; ++t) {
      const int cur = (t - t_begin) & 1;
      const bool has_next = t + 1 < t_end;
      auto cycle = [&](auto cc) __attribute__((always_inline)) {
        constexpr int CYC = decltype(cc)::value;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, CYC * KPC + KPC - 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (CYC == CPT - 1) wg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                // b1
        if (CYC == CPT - 1) {
          flip_buffers(cur == 0 ? BUF : -BUF);       // next tile's first fragments come from the other buffer
          read_frags(std::integral_constant<int, KPC - 1>{});
        } else {
          read_frags(std::integral_constant<int, (CYC + 1) * KPC + KPC - 1 < TH ? (CYC + 1) * KPC + KPC - 1 : 0>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
          if (CYC == 0) dma_setup(t + 1);
          if (CYC < CPT - 1) dma_issue((cur ^ 1) * BUF, std::integral_constant<int, CYC * PER>{}, std::integral_constant<int, (CYC + 1) * PER>{});
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                // b2
      };
      cycle(I0{});
      if constexpr (CPT > 1) cycle(I1{});
      if constexpr (CPT > 2) { cycle(I2{}); cycle(I3{}); }
    }
  }

  // ---- epilogue: lane holds dW[tap][ci = ci0 + wci*16 + g*4 + r][co = co0 + wco*64 + o*16 + (lane & 15)]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + wci * 16 + g * 4 + r;
      if (ci >= a.cin_real) continue;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const int co = co0 + wco * 64 + o * 16 + (lane & 15);
        if (co < a.Cout) atomicAdd(a.dw + ((size_t)(t * a.cin_real + ci) * a.Cout + co), acc[t][o][r]);
      }
    }
  if (do_bias) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      float s = dbsum[o];                             // lanes l, l+16, l+32, l+48 hold the four k-groups of column l & 15
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const int co = co0 + wco * 64 + o * 16 + (lane & 15);
      if (lane < 16 && co < a.Cout) atomicAdd(a.db + co, s);
    }
  }
}

int wg_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

template <int COT, bool FRONT = false>
int launch_wg_halo(WgHaloArgs& a, hipStream_t s) {
  constexpr int XB = 6 * 5 * 1024, YB = 128 * COT * 2;
  constexpr int LDS = 2 * (XB + YB);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_halo_kernel<COT, FRONT>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  a.ci_tiles = a.C / 64;
  a.co_tiles = (a.Co8 + COT - 1) / COT;
  const int pairs = a.ci_tiles * a.co_tiles;
  int splits = wg_cu_count() / pairs;
  if (splits < 1) splits = 1;
  if (splits > a.total_tiles) splits = a.total_tiles;
  a.tiles_per_split = (a.total_tiles + splits - 1) / splits;
  splits = (a.total_tiles + a.tiles_per_split - 1) / a.tiles_per_split;
  a.div_ci = make_fastdiv(a.ci_tiles);
  a.div_pairs = make_fastdiv(pairs);
  a.xcd_grouped = (splits % 8 == 0 && pairs > 1) ? 1 : 0;
  hipLaunchKernelGGL((conv_wgrad_halo_kernel<COT, FRONT>), dim3(pairs * splits), dim3(512), LDS, s, a);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

}  // namespace

static bool wg_halo_eligible(const danhip_conv_desc* d) {
  if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W)) return false;      // 'same' 3x3 only
  const int co8 = (d->Cout + 7) / 8 * 8;
  if (d->Cin % 64 != 0 || (co8 % 64 != 0 && co8 > 64)) return false;      // thin heads (co8 < 64) run as one zero-padded 64-wide tile
  const int th = 4, tw = 32;
  const double util = (double)d->H * d->W / ((double)((d->H + th - 1) / th * th) * (double)((d->W + tw - 1) / tw * tw));
  return util >= 0.6;       // 40x40 / 20x20 maps (0.625) still beat the per-tap kernel: 579 vs 440 TFLOP/s on conv5_1
}

const char* danhip_wgrad_halo_label(const danhip_conv_desc* d) {
  if (!wg_halo_eligible(d)) return nullptr;
  return ((d->Cout + 7) / 8 * 8) % 128 == 0 ? "conv_wgrad_halo_kernel<128, false>" : "conv_wgrad_halo_kernel<64, false>";
}

// Returns DANHIP_OK when launched, 1 when the shape is not eligible (caller falls back to conv_wgrad.hip).
int danhip_launch_wgrad_halo(const danhip_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* db, int cin_real, hipStream_t s) {
  if (!wg_halo_eligible(d)) return 1;
  const int co8 = (d->Cout + 7) / 8 * 8;
  const int th = 4, tw = 32;
  WgHaloArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.db = db;
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->Cin; a.Co8 = co8; a.Cout = d->Cout; a.cin_real = cin_real;
  a.tiles_x = (d->W + tw - 1) / tw;
  a.tiles_y = (d->H + th - 1) / th;
  a.total_tiles = d->N * a.tiles_x * a.tiles_y;
  a.div_tx = make_fastdiv(a.tiles_x);
  a.div_txy = make_fastdiv(a.tiles_x * a.tiles_y);
  return co8 % 128 == 0 ? launch_wg_halo<128>(a, s) : launch_wg_halo<64>(a, s);
}
