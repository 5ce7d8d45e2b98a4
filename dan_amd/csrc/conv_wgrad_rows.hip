// Weight gradient of 3x3 / stride-1 'same' convolutions on MFMA for gfx950 — ROW-STREAMING form.
//
//   dW[tap][ci][co] = sum over pixels  X[pixel + tap][ci] * dY[pixel][co]         (tf.layers.conv2d backward w.r.t. the
//   kernel variable of net/sfd_net.py:81-89 and every 3x3 conv of net/*.py; bias gradient db[co] = sum dY rides along)
//
// The reduction runs over PIXELS.  A K-step is one row of 32 pixels of a 32-pixel-wide column STRIP of one image; a persistent
// 512-thread workgroup owns a (64 ci) x (COT co) x (9 taps) gradient tile in registers (wave tile 16 ci x COT/2 co x 9 taps)
// and walks down its share of strips row by row.  What changed against the tile form (conv_wgrad_halo.hip, round 1):
//
//  * the nine taps of K-step y need X rows y-1, y, y+1 at three column shifts — rows y and y+1 are the rows y+1 and y+2 of
//    the previous K-step.  Each wave therefore keeps a 3-row x 3-shift WINDOW of X fragments in registers (36 VGPRs) and reads
//    only the new row's 3 fragments per K-step: 6 + 2*NO transposing LDS reads per 9*NO MFMAs (0.39 per MFMA at COT = 128; the
//    tile form re-read all 9 fragments: 0.72);
//  * rows stream through two LDS rings (6 X rows of 40 px x 64 ci, 6 dY rows of 32 px x COT co): every X row is fetched ONCE per
//    strip (the tile form fetched a 6-row patch per 4 output rows: 1.5x), 2 LDS-DMA instructions per wave and K-step, four K-steps
//    ahead, retired by a counted vmcnt — never a drain in the loop;
//  * the loop body is six K-steps with every ring slot / window register a compile-time constant (ring depth 6 = lcm with the
//    3-deep window), so the per-step address arithmetic is the DMA's ~20 instructions and nothing else.
//
// Two wave groups alternate phases as in conv_halo.hip: while one group issues its 9*NO MFMAs with nothing else in its stream,
// the other issues its DMA (geometry computed one phase earlier), reads its fragments and advances the scalar row generator; two
// barriers per K-step swap the roles.  The LDS-DMA is inline asm so that hipcc does not drain it (vmcnt(0)) before every ds_read.
// Measured (conv3_2, batch 16, profiles/r2): MFMA busy 48 % -> 64 % of SIMD cycles, LDS-array cycles halved, 0 bank conflicts; s_setprio
// for the memory phase and a DMA-last order were tried and are neutral.
// A strip segment starts with two warm-up K-steps that only feed the window (no MFMA); out-of-image rows / columns are zero-filled
// by the buffer descriptor's range check ('same' padding).  Partial gradients are combined with fp32 atomics into the HWIO tensor
// (zeroed by the caller once per step).
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace {

struct WgRowsArgs {
  const bf16_t* x;     // [N,H,W,C]
  const bf16_t* dy;    // [N,H,W,Co8]
  float* dw;           // [3,3,cin_real,Cout]
  float* db;           // [Cout] or null
  int N, H, W, C, Co8, Cout, cin_real;
  int ldx, ldy;        // pixel pitches of x / dy in elements (C / Co8 when dense; channel-slice views of wider tensors otherwise - round 4)
  int tiles_x, total_rows, rows_per_split;
  int ci_tiles, co_tiles, xcd_grouped;
  int ablate;          // timing experiments only (DANHIP_WGRAD_ABLATE=1): skip the epilogue's atomics
  int b2;              // 1: second barrier per K-step (option "wgrad_b2")
  float* slab;         // optional workspace: every block stores its partial tile here (plain 16-byte stores) and wg_rows_reduce_kernel combines
  int splits;
  FastDiv div_tx, div_h, div_ci, div_pairs;
#ifdef WR_TRACE
  unsigned* trace;     // tools/halo2_trace.hip -DTRACE_WGRAD: [2 groups][128 K-steps][4 stamps] shader clocks of workgroup 0, waves 0 and 4
#endif
#ifdef WR_CLOCK
  unsigned long long* clk;   // tools/clock_probe.hip: [blocks][8] = (s_memtime, s_memrealtime) at kernel entry / loop start / loop end / kernel end
#endif
};

#ifdef WR_CLOCK
#define WR_CLK(slot)                                                                                                   \
  do {                                                                                                                 \
    if (tid == 0) {                                                                                                    \
      a.clk[(size_t)blockIdx.x * 8 + 2 * (slot)] = __builtin_amdgcn_s_memtime();                                       \
      a.clk[(size_t)blockIdx.x * 8 + 2 * (slot) + 1] = __builtin_amdgcn_s_memrealtime();                               \
    }                                                                                                                  \
  } while (0)
#else
#define WR_CLK(slot) do { } while (0)
#endif

#ifdef WR_TRACE
// all-scalar time stamp into LDS behind the two rings (the dynamic LDS segment starts at LDS address 0)
#define WR_STAMP(slot)                                                                                              \
  do {                                                                                                              \
    if (blockIdx.x == 0 && (wave & 3) == 0 && tr_step < 128) {                                                       \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                                                   \
      const unsigned sa_ = (unsigned)(XBASE + DEPTH * XS + (((wave >> 2) * 128 + tr_step) * 4 + (slot)) * 4);       \
      unsigned va_, vd_;                                                                                            \
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_write_b32 %0, %1" : "=&v"(va_), "=&v"(vd_) : "s"(sa_), "s"(t_) : "memory"); \
    }                                                                                                               \
  } while (0)
#define WR_STEP_DONE() (++tr_step)
#else
#define WR_STAMP(slot) do { } while (0)
#define WR_STEP_DONE() do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wr_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// 16-byte-per-lane LDS-DMA as inline asm: hipcc counts a builtin LDS-DMA as a pending LDS write and drains it (vmcnt(0)) in front of
// the next ds_read it cannot prove disjoint — every K-step here.  Hidden in asm the DMA is counted by hand (wr_wait_vmcnt) and stays
// in flight across the barriers.  M0 (the LDS destination base) is written in the statement that uses it and restored after.
typedef __attribute__((ext_vector_type(4))) unsigned wr_u32x4;
__device__ __forceinline__ void wr_dma16(wr_u32x4 rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ wr_u32x4 wr_make_rsrc(const void* p, unsigned bytes) {   // raw buffer descriptor: base, stride 0, num_records, flags
  const unsigned long long a = (unsigned long long)p;
  return wr_u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ int wr_f128(int pc) { return ((pc >> 1) & 1) | (((pc >> 3) & 1) << 1); }      // 32-byte-slot swizzles
__device__ __forceinline__ int wr_f256(int px) { return (px & 3) | (((px >> 3) & 1) << 2); }

// scalar generator of the block's K-step stream (one instance runs ahead for the DMA)
struct RowGen {
  int strip, n, x0, y, warm, left;
};
struct RowStep {
  int n, x0, xrow, yrow;      // xrow / yrow < 0 or >= H: nothing to load (zero fill)
  bool xvalid, yvalid;
};

template <int COT>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_rows_kernel(const WgRowsArgs a) {
  constexpr int TW = 32;
  constexpr int PITCH = 40;                           // X ring row: 34 pixels used, 5 DMA pieces of 8 pixels x 128 bytes
  constexpr int XS = PITCH * 128;                     // 5120 bytes per X row
  constexpr int RBY = COT * 2;                        // dY pixel bytes
  constexpr int YS = TW * RBY;                        // bytes per dY row (8192 / 4096)
  constexpr int DEPTH = 6;                            // ring depth (both rings) = unroll of the loop body
  constexpr int P = 4;                                // K-steps the DMA runs ahead
  constexpr int YBASE = 0, XBASE = DEPTH * YS;
  constexpr int NO = COT / 32;                        // co fragments per wave (64 / 32 co)
  constexpr int YPIECES = YS / 1024;                  // 8 / 4
  static_assert(XBASE + DEPTH * XS <= 160 * 1024, "LDS budget");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave & 3;
  const int grp = wave >> 2;                          // phase group (0 = A, 1 = B) and co half
  const int wco = grp;
  WR_CLK(0);                                          // (diagnostic build only; the store retires under the prologue's vmcnt(0))

  // ---- block -> (ci tile, co tile, split of the K-step stream); the (ci, co) pairs of one split read the same rows: same XCD
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
  const int k_begin = split * a.rows_per_split;
  const int k_end = min(a.total_rows, k_begin + a.rows_per_split);
  if (k_begin >= k_end) return;
  // K-steps of the block: its rows + two warm-up steps per strip segment, padded to the unroll
  const int strip_first = (int)fdiv((unsigned)k_begin, a.div_h), strip_last = (int)fdiv((unsigned)(k_end - 1), a.div_h);
  const int steps = (k_end - k_begin) + 2 * (strip_last - strip_first + 1);
  const int V = (steps + DEPTH - 1) / DEPTH * DEPTH;

  // (the last pixel's row ends C / Co8 channels after its start whatever the pitch)
  const wr_u32x4 rsrc_x = wr_make_rsrc(a.x, ((unsigned)(a.N * a.H * a.W - 1) * (unsigned)a.ldx + (unsigned)a.C) * 2u);
  const wr_u32x4 rsrc_y = wr_make_rsrc(a.dy, ((unsigned)(a.N * a.H * a.W - 1) * (unsigned)a.ldy + (unsigned)a.Co8) * 2u);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS char*)smem);

  // ---- the K-step generator (wave-uniform scalars)
  RowGen gen;
  auto gen_strip = [&](RowGen& g) __attribute__((always_inline)) {
    g.n = (int)fdiv((unsigned)g.strip, a.div_tx);
    g.x0 = (g.strip - g.n * a.tiles_x) * TW;
  };
  gen.strip = strip_first;
  gen.y = k_begin - strip_first * a.H;
  gen.warm = 2;
  gen.left = k_end - k_begin;
  gen_strip(gen);
  unsigned long long flags = 0;                       // bit i: K-step (current + i) has MFMAs
  auto gen_next = [&](RowStep& s, int bit) __attribute__((always_inline)) {
    s.n = gen.n; s.x0 = gen.x0;
    if (gen.left == 0) {
      s.xvalid = false; s.yvalid = false; s.xrow = 0; s.yrow = 0;
    } else if (gen.warm > 0) {
      s.xrow = gen.y - gen.warm + 1; s.xvalid = true; s.yvalid = false; s.yrow = 0;
      gen.warm -= 1;
    } else {
      s.xrow = gen.y + 1; s.yrow = gen.y; s.xvalid = true; s.yvalid = true;
      flags |= 1ull << bit;
      gen.y += 1; gen.left -= 1;
      if (gen.y == a.H) { gen.strip += 1; gen.y = 0; gen.warm = 2; gen_strip(gen); }
    }
  };

  // ---- DMA of one K-step into ring slot SL: every wave issues one X piece and one dY piece.  A lane's part of the source offset
  // (pixel in the piece, swizzled chunk) never changes; its column validity changes with the strip only: both are kept in registers
  // and a K-step adds the scalar row base — 2 VALU per DMA (the tile kernel spent ~20).
  const int pc5 = wave < 4 ? wave : 4;               // X piece of this wave; waves 5..7 repeat piece 4 (same bytes, same place)
  constexpr int CPP = RBY / 16;                       // dY: 16-byte chunks per pixel (16 / 8)
  constexpr int PXP = 1024 / RBY;                     // dY: pixels per piece (4 / 8)
  const int ypiece = wave & (YPIECES - 1);            // COT = 64: waves 4..7 repeat pieces 0..3
  const int l8 = lane >> 3, lp = lane / CPP;
  unsigned vx, vy;                                    // lane offsets
  {
    const int xA = (lane & 7) ^ (((l8 >> 1) & 1) << 1);
    const int chunk = xA ^ ((pc5 & 1) << 2);          // = (lane & 7) ^ (f128(pcol) << 1), pcol = pc5*8 + l8
    vx = __umul24((unsigned)l8, (unsigned)(a.ldx * 2)) + (unsigned)(chunk << 4);
    const int cpos = lane % CPP, pxb = ypiece * PXP;
    const int yB = COT == 128 ? (cpos ^ (lp << 1)) : (cpos ^ (((lp >> 1) & 1) << 1));
    const int sb = (pxb >> 3) & 1;
    const int ychunk = yB ^ (COT == 128 ? (sb << 3) : (sb << 2));       // = cpos ^ (f256 / f128 (pxb + lp) << 1)
    vy = __umul24((unsigned)lp, (unsigned)(a.ldy * 2)) + (unsigned)(ychunk << 4);
    if (!(co0 + ychunk * 8 < a.Co8)) vy = 0xFFFFFFFFu;                   // thin heads: channel chunks beyond Co8 are zero-filled
  }
  unsigned vxbad = 0, vybad = 0;                      // column validity of the current strip (all ones = zero fill)
  int dma_x0 = -1;
  auto dma_step = [&](auto slc, const RowStep& s) __attribute__((always_inline)) {
    constexpr int SL = decltype(slc)::value;
    if (s.x0 != dma_x0) {                             // new strip (wave-uniform, once per ~H K-steps)
      dma_x0 = s.x0;
      const int x = s.x0 - 1 + pc5 * 8 + l8;
      vxbad = (l8 < (pc5 == 4 ? 2 : 8) && (unsigned)x < (unsigned)a.W) ? 0u : 0xFFFFFFFFu;          // 34 pixels
      vybad = (s.x0 + ypiece * PXP + lp < a.W) ? 0u : 0xFFFFFFFFu;
    }
    {
      const int xs = s.x0 - 1 + pc5 * 8;
      const unsigned bady = (s.xvalid && (unsigned)s.xrow < (unsigned)a.H) ? 0u : 0xFFFFFFFFu;
      const unsigned sbase = (unsigned)(((s.n * a.H + s.xrow) * a.W + xs) * a.ldx + ci0) * 2u;      // may wrap; exact for valid lanes
      const unsigned voff = (vx + sbase) | vxbad | bady;
      wr_dma16(rsrc_x, voff, lds0 + XBASE + SL * XS + pc5 * 1024);
    }
    {
      const int xs = s.x0 + ypiece * PXP;
      const unsigned bady = (s.yvalid && (unsigned)s.yrow < (unsigned)a.H) ? 0u : 0xFFFFFFFFu;
      const unsigned sbase = (unsigned)(((s.n * a.H + s.yrow) * a.W + xs) * a.ldy + co0) * 2u;
      const unsigned voff = (vy + sbase) | vybad | bady;
      wr_dma16(rsrc_y, voff, lds0 + YBASE + SL * YS + ypiece * 1024);
    }
  };

  // ---- fragment addresses of ring slot 0 (other slots: immediate offsets)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  int xaddr[3][2];                                    // [column shift j][half h]
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pcol = j + 8 * g + 4 * h + q;
      const int ch = (wci * 2 + (p >> 1)) ^ (wr_f128(pcol) << 1);
      xaddr[j][h] = XBASE + pcol * 128 + (ch << 4) + (p & 1) * 8;
    }
  int yaddr[NO];                                      // half 1 (pixel + 4) is + 4*RBY: the swizzle does not see bit 2 of the pixel
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const int px = 8 * g + q;
    const int ch = ((wco * NO + o) * 2 + (p >> 1)) ^ ((COT == 128 ? wr_f256(px) : wr_f128(px)) << 1);
    yaddr[o] = YBASE + px * RBY + (ch << 4) + (p & 1) * 8;
  }

  f32x4 acc[9][NO];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[t][o] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient db[co] = sum over pixels dY[.][co]: one extra MFMA with an all-ones A fragment (every row of the 16 x 16 result is
  // the column sum).  The blocks of ci tile 0 carry it, wave (wci, wco) for co fragment o = wci of its half: 1 MFMA in 9*NO + 1.
  const bool do_bias = a.db != nullptr && ci_tile == 0 && wci < NO;     // wave-uniform
  f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 win[3][3], yf[NO];                           // win[window slot][column shift]
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < 3; ++j) win[s][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  auto tr2 = [&](int addr) __attribute__((always_inline)) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(smem + addr));
    return lo;
  };
  auto mem = [&](auto uc) __attribute__((always_inline)) {              // fragments of the K-step in ring slot U
    constexpr int U = decltype(uc)::value;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const s16x4 lo = tr2(yaddr[o] + U * YS), hi = tr2(yaddr[o] + 4 * RBY + U * YS);
      yf[o] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const s16x4 lo = tr2(xaddr[j][0] + U * XS), hi = tr2(xaddr[j][1] + U * XS);
      win[U % 3][j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  auto mma = [&](auto uc) __attribute__((always_inline)) {              // newest window row is slot U % 3 = tap row 2
    constexpr int U = decltype(uc)::value;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[i * 3 + j][o] = DH_MFMA_16x16x32(win[(U + 1 + i) % 3][j], yf[o], acc[i * 3 + j][o]);
    if (do_bias) {
      const act16_t one = (act16_t)1.0f;
      const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (wci == o) {                                 // scalar branch per arm (a select over yf[] would become a scratch array)
          accb = DH_MFMA_16x16x32(ones, yf[o], accb);
          asm volatile("" : "+v"(accb));
        }
    }
  };

  // ---- prologue: K-steps 0 .. P-1 into slots 0 .. P-1
  {
    RowStep s;
    gen_next(s, 0); dma_step(std::integral_constant<int, 0>{}, s);
    gen_next(s, 1); dma_step(std::integral_constant<int, 1>{}, s);
    gen_next(s, 2); dma_step(std::integral_constant<int, 2>{}, s);
    gen_next(s, 3); dma_step(std::integral_constant<int, 3>{}, s);
    static_assert(P == 4, "prologue issues P steps");
  }
  WR_CLK(1);
  wr_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // K-step v (ring slot U = v % 6):   A:  mem(v) + DMA(v+P) | b1 | MFMA(v) | b2        B:  MFMA(v) | b1 | mem(v+1) + DMA(v+1+P) | b2
  // A wave's pieces of K-step u are waited for (vmcnt(2(P-1)): the P-1 younger steps stay in flight) before b1 of cycle u-1; B reads
  // them in the phase after that barrier, A one phase later.  b2 is off by default (option "wgrad_b2"; -2.5 % on the layer set): a slot is
  // rewritten only by waves that have passed the b1 after its last readers' reads completed (argument in conv_halo.hip's main loop).
  RowStep nxt;
  [[maybe_unused]] int tr_step = 0;
  if (grp == 0) {
    gen_next(nxt, P);
    for (int v = 0; v < V; v += DEPTH) {
      auto step = [&](auto uc) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        WR_STAMP(0);
        dma_step(std::integral_constant<int, (U + P) % DEPTH>{}, nxt);   // geometry from the previous phase
        __builtin_amdgcn_sched_barrier(0);
        mem(uc);
        __builtin_amdgcn_sched_barrier(0);
        gen_next(nxt, P + 1);                        // the next K-step's DMA geometry (scalar work under the reads' latency)
        wr_wait_vmcnt<2 * (P - 1)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        WR_STAMP(1);
        __builtin_amdgcn_s_barrier();                // b1
        WR_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        if (flags & 1ull) mma(uc);
        flags >>= 1;
        __builtin_amdgcn_sched_barrier(0);
        WR_STAMP(3);
        if (a.b2) __builtin_amdgcn_s_barrier();      // b2
        WR_STEP_DONE();
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    }
  } else {
    {
      mem(std::integral_constant<int, 0>{});
      gen_next(nxt, P);
      dma_step(std::integral_constant<int, P % DEPTH>{}, nxt);
      gen_next(nxt, P + 1);
    }
    for (int v = 0; v < V; v += DEPTH) {
      auto step = [&](auto uc) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        WR_STAMP(0);
        __builtin_amdgcn_sched_barrier(0);
        if (flags & 1ull) mma(uc);
        flags >>= 1;
        __builtin_amdgcn_sched_barrier(0);
        WR_STAMP(1);
        wr_wait_vmcnt<2 * (P - 1)>();
        __builtin_amdgcn_s_barrier();                // b1
        WR_STAMP(2);
        dma_step(std::integral_constant<int, (U + 1 + P) % DEPTH>{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
        mem(std::integral_constant<int, (U + 1) % DEPTH>{});
        __builtin_amdgcn_sched_barrier(0);
        gen_next(nxt, P + 1);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        WR_STAMP(3);
        if (a.b2) __builtin_amdgcn_s_barrier();      // b2
        WR_STEP_DONE();
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    }
  }
  wr_wait_vmcnt<0>();                                 // zero-fill pieces of the steps beyond the stream are still landing
  WR_CLK(2);
#ifdef WR_TRACE
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = tid; i < 1024; i += 512) a.trace[i] = reinterpret_cast<const unsigned*>(smem + XBASE + DEPTH * XS)[i];
#endif

  if (a.ablate) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int o = 0; o < NO; ++o) asm volatile("" ::"v"(acc[t][o]));
    WR_CLK(3);
    return;
  }
  // ---- epilogue: lane holds dW[tap][ci = ci0 + wci*16 + g*4 + r][co = co0 + wco*NO*16 + o*16 + (lane & 15)]
  // Slab form (a.slab): the block's partial tile goes out as 9 * NO fully coalesced 1 KiB-per-wave stores in register order
  // ([t * NO + o][wave][lane] float4) and wg_rows_reduce_kernel sums the splits and applies the HWIO permutation.  Plain stores run at
  // ~6 TB/s; the same bytes as 256-byte float atomics at ~1.3 TB/s (MI355X_MICROARCH.md): with every block ending at the same time the
  // atomic tail was 50 us of a 100 us launch at 2 images per GPU (profiles/r3/README.md).
  if (a.slab) {
    f32x4* dst = reinterpret_cast<f32x4*>(a.slab) + (size_t)blockIdx.x * (9 * NO * 512) + wave * 64 + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int o = 0; o < NO; ++o) __builtin_nontemporal_store(acc[t][o], dst + (t * NO + o) * 512);
    if (do_bias) {
      const int co = co0 + wco * NO * 16 + wci * 16 + (lane & 15);
      if (lane < 16 && co < a.Cout) atomicAdd(a.db + co, accb[0]);
    }
    WR_CLK(3);
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + wci * 16 + g * 4 + r;
      if (ci >= a.cin_real) continue;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const int co = co0 + wco * NO * 16 + o * 16 + (lane & 15);
        if (co < a.Cout) atomicAdd(a.dw + ((size_t)(t * a.cin_real + ci) * a.Cout + co), acc[t][o][r]);
      }
    }
  if (do_bias) {                                      // rows 0..15 of accb are identical: lanes 0..15 (row 0) deliver
    const int co = co0 + wco * NO * 16 + wci * 16 + (lane & 15);
    if (lane < 16 && co < a.Cout) atomicAdd(a.db + co, accb[0]);
  }
#ifdef WR_CLOCK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the epilogue's atomics have been accepted by the memory system when the end stamp is taken
  WR_CLK(3);
#endif
}

// Second pass of the slab form: a 256-thread workgroup owns 64 consecutive float4 slots of one (ci, co) tile; wave w sums the splits
// w, w + 4, ... (eight independent 16-byte loads in flight per lane), the four partial sums meet in LDS and wave 0 adds the result (four
// consecutive ci of one co) into the HWIO gradient.  Slot -> element as in the kernel above: q = (t * NO + o) * 512 + wave * 64 + lane.
template <int COT>
__global__ __launch_bounds__(256) void wg_rows_reduce_kernel(const WgRowsArgs a) {
  constexpr int NO = COT / 32, TILE = 9 * NO * 512;
  __shared__ f32x4 part[4][64];
  const int pairs = a.ci_tiles * a.co_tiles;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int chunk = blockIdx.x;                       // 64-slot chunk; TILE % 64 == 0, so a chunk never straddles two tiles
  const int pair = chunk / (TILE / 64), q = (chunk - pair * (TILE / 64)) * 64 + lane;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.slab) + q;
  auto blk = [&](int split) -> size_t {             // the block that computed (split, pair): inverse of the kernel's mapping
    return a.xcd_grouped ? (size_t)(((split >> 3) * pairs + pair) * 8 + (split & 7)) : (size_t)(split * pairs + pair);
  };
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int sp = w;
  for (; sp + 28 < a.splits; sp += 32) {              // eight splits of this wave per round
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
  const int to = q >> 9, wave = (q >> 6) & 7;
  const int t = to / NO, o = to - t * NO;
  const int wci = wave & 3, wco = wave >> 2, g = lane >> 4;
  const int co_tile = pair / a.ci_tiles, ci_tile = pair - co_tile * a.ci_tiles;
  const int co = co_tile * COT + wco * NO * 16 + o * 16 + (lane & 15);
  if (co >= a.Cout) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ci = ci_tile * 64 + wci * 16 + g * 4 + r;
    if (ci < a.cin_real) a.dw[(size_t)(t * a.cin_real + ci) * a.Cout + co] += sum[r];
  }
}

#ifdef WR_CLOCK
static int g_wr_clock_blocks = 0, g_wr_clock_steps = 0;
unsigned long long* wr_clock_buffer() {
  static unsigned long long* p = [] { void* q = nullptr; (void)hipMalloc(&q, 4096 * 64); (void)hipMemset(q, 0, 4096 * 64); return (unsigned long long*)q; }();
  return p;
}
int wr_clock_blocks() { return g_wr_clock_blocks; }
int wr_clock_steps() { return g_wr_clock_steps; }
#endif

#ifdef WR_TRACE
unsigned* wr_trace_buffer() {
  static unsigned* p = [] { void* q = nullptr; (void)hipMalloc(&q, 4096 + 64); (void)hipMemset(q, 0, 4096 + 64); return (unsigned*)q; }();
  return p;
}
#endif

int wr_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

template <int COT>
int launch_wg_rows(WgRowsArgs& a, hipStream_t s) {
#ifdef WR_TRACE
  constexpr int LDS = 6 * (32 * COT * 2) + 6 * 40 * 128 + 4096;
  a.trace = wr_trace_buffer();
#else
  constexpr int LDS = 6 * (32 * COT * 2) + 6 * 40 * 128;
#endif
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rows_kernel<COT>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  a.ci_tiles = a.C / 64;
  a.co_tiles = (a.Co8 + COT - 1) / COT;
  const int pairs = a.ci_tiles * a.co_tiles;
  int splits = wr_cu_count() / pairs;
  if (splits < 1) splits = 1;
  if (splits > a.total_rows) splits = a.total_rows;
  a.rows_per_split = (a.total_rows + splits - 1) / splits;
  splits = (a.total_rows + a.rows_per_split - 1) / a.rows_per_split;
  a.div_ci = make_fastdiv(a.ci_tiles);
  a.div_pairs = make_fastdiv(pairs);
  a.xcd_grouped = (splits % 8 == 0 && pairs > 1) ? 1 : 0;
  static const int ablate = [] { const char* e = getenv("DANHIP_WGRAD_ABLATE"); return e ? atoi(e) : 0; }();
  a.ablate = ablate;
  a.b2 = danhip_option("wgrad_b2");
  a.splits = splits;
  // The slab form pays when the launch is short (every block reaches its epilogue together and nothing hides the tail: 2-4 images per
  // GPU, the 40x40 / 20x20 levels); on long launches the blocks drift apart, the atomic tail hides under other blocks' MFMAs and the
  // extra pass costs more than it saves (batch 16: conv3_2 0.426 against 0.428 ms, conv2_2 0.479 against 0.455; profiles/r3).
  const int slab_mode = danhip_option("wgrad_slab");
  if (a.slab && (splits < 2 || (slab_mode != 2 && a.rows_per_split > 192))) a.slab = nullptr;
#ifdef WR_CLOCK
  a.clk = wr_clock_buffer();
  g_wr_clock_blocks = pairs * splits;
  g_wr_clock_steps = a.rows_per_split;
#endif
  hipLaunchKernelGGL((conv_wgrad_rows_kernel<COT>), dim3(pairs * splits), dim3(512), LDS, s, a);
  DH_LAUNCH_CHECK();
  if (a.slab) {
    const int chunks = pairs * 9 * (COT / 32) * 512 / 64;
    hipLaunchKernelGGL((wg_rows_reduce_kernel<COT>), dim3(chunks), dim3(256), 0, s, a);
    DH_LAUNCH_CHECK();
  }
  return DANHIP_OK;
}

}  // namespace

// output-channel tile: 128 when the 64-padded channel count is a multiple of 128 (72 -> 128: one tile, 56 zero-filled columns)
static int wg_rows_cot(const danhip_conv_desc* d) { return ((d->Cout + 63) / 64 * 64) % 128 == 0 ? 128 : 64; }

static bool wg_rows_eligible(const danhip_conv_desc* d) {
  if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W)) return false;      // 'same' 3x3 only
  if (d->Cin % 64 != 0) return false;      // any Cout: channel chunks beyond Co8 are zero-filled (thin heads: one 64-wide tile; 72 -> one 128-wide tile)
  const int tw = 32;
  const double util = (double)d->W / (double)((d->W + tw - 1) / tw * tw);
  return util >= 0.6;       // 40- and 20-wide maps (0.625) still beat the per-tap kernel
}

const char* danhip_wgrad_rows_label(const danhip_conv_desc* d) {
  if (!wg_rows_eligible(d)) return nullptr;
  return wg_rows_cot(d) == 128 ? "conv_wgrad_rows_kernel<128>" : "conv_wgrad_rows_kernel<64>";
}

// Bytes of the partial-tile workspace the slab form needs for this descriptor (one register tile per workgroup, at most one workgroup
// per CU); 0 when the row-streaming kernel does not take the shape.
size_t danhip_wgrad_rows_workspace_bytes(const danhip_conv_desc* d) {
  if (!wg_rows_eligible(d)) return 0;
  const int cot = wg_rows_cot(d);
  // the launch geometry of launch_wg_rows: the slab form is taken for short launches only (rows_per_split <= 192)
  const int co8 = (d->Cout + 7) / 8 * 8;
  const int pairs = (d->Cin / 64) * ((co8 + cot - 1) / cot);
  const int total_rows = d->N * ((d->W + 31) / 32) * d->H;
  int splits = wr_cu_count() / pairs;
  if (splits < 1) splits = 1;
  if (splits > total_rows) splits = total_rows;
  const int rows_per_split = (total_rows + splits - 1) / splits;
  const int slab_mode = danhip_option("wgrad_slab");
  if (splits < 2 || (slab_mode != 2 && rows_per_split > 192)) return 0;
  return (size_t)wr_cu_count() * 9 * (cot / 32) * 512 * 16;
}

// Returns DANHIP_OK when launched, 1 when the shape is not eligible (caller falls back to conv_wgrad.hip).
int danhip_launch_wgrad_rows(const danhip_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* db, int cin_real, hipStream_t s,
                             void* ws, size_t ws_bytes, int ldx, int ldy) {
  if (!wg_rows_eligible(d)) return 1;
  const int co8 = (d->Cout + 7) / 8 * 8;
  if (ldx > 0xFFFF || ldy > 0xFFFF) return 1;       // (the lane offsets are 24-bit products)
  WgRowsArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.db = db;
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->Cin; a.Co8 = co8; a.Cout = d->Cout; a.cin_real = cin_real;
  a.ldx = ldx ? ldx : d->Cin; a.ldy = ldy ? ldy : co8;
  a.tiles_x = (d->W + 31) / 32;
  a.total_rows = d->N * a.tiles_x * d->H;
  a.div_tx = make_fastdiv(a.tiles_x);
  a.div_h = make_fastdiv(d->H);
  a.slab = (ws && ws_bytes >= danhip_wgrad_rows_workspace_bytes(d)) ? reinterpret_cast<float*>(ws) : nullptr;
  return wg_rows_cot(d) == 128 ? launch_wg_rows<128>(a, s) : launch_wg_rows<64>(a, s);
}
