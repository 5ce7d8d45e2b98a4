// 3x3 / stride-1 convolution with 64 input and 64 output channels (conv1_2 of every backbone, net/sfd_net.py:128, and its
// data gradient; DAN's 64->64 branch conv, net/danet.py:880) — register-resident weights.
//
// At C = Co = 64 the whole filter is 9 x 64 x 64 bf16 = 72 KiB: each of the 8 waves keeps the 32 output channels it owns
// (9 taps x 2 k-slices x 2 channel tiles = 36 MFMA A-fragments = 144 VGPRs) in registers for the lifetime of the persistent
// workgroup.  No weight DMA, no weight LDS reads, and the only workgroup barrier is one per spatial tile (patch hand-off):
// per 8 x 32 pixel tile a wave (64 pixels x 32 channels) reads 72 pixel fragments and issues 144 MFMAs; the next tile's halo
// patch (43 KiB, double buffered) lands meanwhile and the epilogue stores drain under the next tile's MFMAs.
// The kernel is HBM-lean by construction (input read once + halo, output written once) — at 640 x 640 x batch 16 that is
// 1.7 GB per call, so it sits near both rooflines (0.3 ms at 5.5 TB/s; 0.3 ms at 1.6 PFLOP/s).
#include <type_traits>

#include "conv_common.h"

namespace {

struct C64Geom {
  int tiles_x, tiles_y, sp_items;
  FastDiv div_tx, div_txy;
};

template <int N>
__device__ __forceinline__ void c64_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// FUSE8 (round 6, data gradient only): the call is conv1_2's data gradient and its output dX is conv1_1's dY, which nothing reads but conv1_1's
// WEIGHT gradient (the first layer has no data gradient).  Instead of storing dX (839 MB per batch-16 step at 640 x 640) for conv_wgrad_c8.hip to
// read back (188 us, the LAST kernel of backward, with the other queue empty), every tile's masked dX goes to LDS in 16 bits — the values the store
// would have written — and multiplied with the tile's 10 x 34 patch of the 8-channel image as conv_wgrad_c8.hip does (both MFMA operands by
// ds_read_b64_tr_b16, the nine taps as address offsets into the patch, tap slot 9 = the constant [1, 0, 0, 0] whose row is the bias gradient).
// The register file is full (144 weight + 32 accumulator + 48 pixel-fragment registers of 256), so the gradient tile M = 12 tap slots x 4 channels,
// N = 64 is split over the waves by N: wave w owns output channels 16 (w & 3) .. +15 for tile rows 4 (w >> 2) .. +3 — 12 accumulator registers,
// 12 MFMAs and 32 transposing reads per tile, one extra workgroup barrier.  At the end the waves' accumulators are summed in LDS and leave the CU
// as 1728 + 64 fp32 atomics.
template <bool DGRAD, bool FUSE8 = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_c64_kernel(const ConvArgs a, const C64Geom g) {
  static_assert(DGRAD || !FUSE8, "the folded first-layer weight gradient belongs to the data gradient");
  constexpr int TH = 8, TW = 32, PW = TW + 2;
  constexpr int PROWS = (TH + 2) * PW, PPIECES = (PROWS + 7) / 8, PBYTES = PPIECES * 1024, PL = (PPIECES + 7) / 8;
  constexpr int NPT = 4, NCT = 2;                  // wave tile: 64 pixels (2 tile rows) x 32 channels
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int G = gridDim.x;

  // ---- this wave's weights -> registers: A fragment (tap, ks, ct): W[co][k = tap*64 + ks*32 + fq*8 ..] with the rows of the two
  //      channel tiles interleaved, co = wn*32 + (frow>>2)*8 + ct*4 + (frow&3): accumulator rows fq*4..+3 of tiles ct = 0, 1 are then
  //      the 8 CONSECUTIVE channels wn*32 + fq*8 .. +7 of one pixel -> one 16-byte store per lane and pixel
  bf16x8 wr[9][2][NCT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        wr[t][ks][ct] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(wn * 32 + (frow >> 2) * 8 + ct * 4 + (frow & 3)) * a.Kpad + t * 64 + ks * 32 + fq * 8);

  auto sp_coords = [&](int sp, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    n = (int)fdiv((unsigned)sp, g.div_txy);
    const int rem = sp - n * (g.tiles_x * g.tiles_y);
    const int ty = (int)fdiv((unsigned)rem, g.div_tx);
    y0 = ty * TH;
    x0 = (rem - ty * g.tiles_x) * TW;
  };

  // ---- patch DMA (same layout as conv_halo.hip: row R = hy*PW + hx, chunk c at position c ^ (hx & 7))
  // (channel-slice views, round 4: pixel pitches a.ldx / a.ldy / a.ldm instead of the dense 64 - DAN's 64 -> 64 branch convolutions read and
  // write slices of wider tensors; the last pixel's row ends 64 channels after its start whatever the pitch)
  const unsigned npix1 = (unsigned)(a.N * a.H * a.W) - 1u;
  const unsigned px_x = (unsigned)a.ldx * 2u, px_y = (unsigned)a.ldy * 2u, px_m = (unsigned)a.ldm * 2u;      // bytes per pixel
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)(npix1 * px_x + 128u), 0x00020000);
  auto issue_patch = [&](int sp, int buf) __attribute__((always_inline)) {
    int n, y0, x0;
    sp_coords(sp, n, y0, x0);
    int ln = lane;
    asm volatile("" : "+v"(ln));                   // keep the per-piece geometry out of long-lived registers (recomputed per tile)
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      int piece = k * 8 + wave;
      if (piece > PPIECES - 1) piece = PPIECES - 1;
      const int row = piece * 8 + (ln >> 3);
      const int hy = row / PW, hx = row - hy * PW;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = row < PROWS && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)((n * a.H + y) * a.W + x) * px_x + (unsigned)(((ln & 7) ^ (hx & 7)) << 4) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_AS void*)(smem + buf * PBYTES + piece * 1024), 16, off, 0, 0, 0);
    }
  };

  // ---- epilogue inputs through LDS.  A global load in the epilogue is a full memory latency with nothing to hide it (both waves of a
  // SIMD are in the same epilogue).  Forward: the 64 bias floats are staged once.  Data gradient: the ReLU mask and the value to accumulate
  // into are DMA'd at the START of the tile into a region private to the wave that consumes them — 64 pixels x its 32 channels = 4 KiB,
  // lane l of DMA k fetches the 16 bytes that lane (frow = l >> 2, fq = l & 3) of fragment k reads back — land under the tile's 144 MFMAs
  // and are covered by the tile-end vmcnt(0).  No other wave touches the region: no barrier, no double buffer.
  constexpr int RWBASE = 2 * PBYTES;               // LDS map: [patch 0][patch 1][forward: bias | dgrad: 8 x 4 KiB mask, 8 x 4 KiB old]
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_m =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.mask ? a.mask : a.x), 0, (int)(npix1 * (a.mask ? px_m : px_x) + 128u), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_o =
      __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t*>(a.y), 0, (int)(npix1 * px_y + 128u), 0x00020000);
  // Bit-mask form (ConvArgs::mask_bits, danhip_relu_bits layout: 8 bytes per pixel): the tile's 256 pixels x 8 bytes = 2 KiB are fetched by
  // waves 0 and 1 (one 16-byte DMA per lane = two horizontally adjacent pixels) into one of two 2 KiB buffers — the hand-off barrier that
  // ends the tile publishes them to every wave's epilogue, and the buffer is refilled two tiles later, i.e. behind the NEXT hand-off
  // barrier, which every wave passes only after its epilogue reads.  1/16 of the 16-bit mask's bytes (conv1_2, batch 16: 839 -> 52 MB).
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.mask_bits ? a.mask_bits : reinterpret_cast<const unsigned char*>(a.x)), 0, (int)((unsigned)(a.N * a.H * a.W) * 8u),
      0x00020000);
  auto issue_bits = [&](int sp, int bbuf) __attribute__((always_inline)) {
    if (wave >= 2) return;                         // (uniform per wave)
    int n, y0, x0;
    sp_coords(sp, n, y0, x0);
    int ln = lane;
    if constexpr (FUSE8) asm volatile("" : "+v"(ln));      // (the folded form has no register left for hoisted per-lane geometry: it was spilled and reloaded per tile)
    const int t = (wave * 64 + ln) * 2;            // pixels t, t + 1 of the tile (same row: TW is even)
    const int y = y0 + t / TW, x = x0 + t % TW;
    const unsigned off = (y < a.H && x < a.W) ? (unsigned)((n * a.H + y) * a.W + x) * 8u : 0xFFFFFFFFu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (LDS_AS void*)(smem + RWBASE + bbuf * 2048 + wave * 1024), 16, off, 0, 0, 0);
  };
  // FUSE8 LDS map inside the epilogue-input region (bit-mask form only: the 16-bit mask / old-value areas are unused):
  //   [RWBASE, +4096) bit masks (2 buffers) | [+4096, +4096 + 2 * 6144) image patches (2 buffers: 340 pixels x 16 bytes in six 1 KiB DMA pieces)
  //   | [+16384, +16384 + 32768) the dX tile [256 pixels][64 channels] | [+49152, +16) the constant fragment [1, 0, 0, 0 | 0 ...]
  //   | [+49216, +49216 + 24576) the waves' gradient accumulators between tiles: [wave][M tile][lane] x 16 bytes.  The register file has no
  //   room for them across the tile's 144 MFMAs (the compiler parked them in scratch and reloaded them, a memory latency per tile, right in
  //   front of the MFMAs that need them: 615 us per launch against 476 for the plain data gradient); from LDS the reload is issued before
  //   the tile-written barrier and has landed when the barrier opens.
  constexpr int IMGBASE = RWBASE + 4096, IMGBYTES = 6144, DXBASE = RWBASE + 16384, C1BASE = RWBASE + 49152, FACCBASE = RWBASE + 49216;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_i = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(FUSE8 ? a.fuse_x8 : a.x), 0, (int)((unsigned)(a.N * a.H * a.W) * 16u), 0x00020000);
  [[maybe_unused]] auto issue_img = [&](int sp, int ibuf) __attribute__((always_inline)) {
    if (wave >= 6) return;                         // (uniform per wave) 340 patch pixels = six pieces of 64
    int n, y0, x0;
    sp_coords(sp, n, y0, x0);
    int ln = lane;
    asm volatile("" : "+v"(ln));                   // (recomputed per tile: the register file has no room for hoisted geometry)
    const int e = wave * 64 + ln;
    const int r = e / PW, c = e - r * PW;
    const int y = y0 - 1 + r, x = x0 - 1 + c;
    const bool ok = e < PROWS && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_i, (LDS_AS void*)(smem + IMGBASE + ibuf * IMGBYTES + wave * 1024), 16,
                                             ok ? (unsigned)((n * a.H + y) * a.W + x) * 16u : 0xFFFFFFFFu, 0, 0, 0);
  };
  auto issue_rw = [&](int sp) __attribute__((always_inline)) {
    int n, y0, x0;
    sp_coords(sp, n, y0, x0);
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int t = wm * 64 + k * 16 + (ln >> 2);
      const int y = y0 + t / TW, x = x0 + t % TW;
      const bool ok = y < a.H && x < a.W;
      const unsigned pix = (unsigned)((n * a.H + y) * a.W + x), ch = (unsigned)((wn * 4 + (ln & 3)) << 4);
      if (a.mask)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_m, (LDS_AS void*)(smem + RWBASE + wave * 4096 + k * 1024), 16, ok ? pix * px_m + ch : 0xFFFFFFFFu, 0, 0, 0);
      if (a.accumulate)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_o, (LDS_AS void*)(smem + RWBASE + 32768 + wave * 4096 + k * 1024), 16, ok ? pix * px_y + ch : 0xFFFFFFFFu,
                                                 0, 0, 0);
    }
  };
  const int rwaddr = RWBASE + wave * 4096 + ((frow * 4 + fq) << 4);     // + p * 1024 (+ 32768: old value)

  int pxaddr[3][NPT];                              // fragment p at tap column j, k-slice 0, in the CURRENT buffer (flipped per tile)
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int t = wm * 64 + p * 16 + frow;
    const int ty = t / TW, tx = t % TW;
#pragma unroll
    for (int j = 0; j < 3; ++j) pxaddr[j][p] = (ty * PW + tx + j) * 128 + ((fq ^ ((tx + j) & 7)) << 4);
  }

  f32x4 acc[NCT][NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int p = 0; p < NPT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  // FUSE8: this wave's slice of conv1_1's gradient (12 tap slots x 4 channels x its 16 output channels), accumulated over all its tiles — in LDS
  [[maybe_unused]] const int faddr = FACCBASE + (wave * 3 * 64 + lane) * 16;       // + mt * 1024
  if constexpr (FUSE8) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) *reinterpret_cast<f32x4*>(smem + faddr + mt * 1024) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 16) reinterpret_cast<bf16_t*>(smem + C1BASE)[tid] = tid == 0 ? f2bf(1.0f) : (bf16_t)0;
  }

  bf16x8 xa[NPT], xb[NPT], xc[NPT];
  auto read_px = [&](bf16x8 (&x)[NPT], auto tapc, int ks) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr int TI = TAP / 3, TJ = TAP % 3;
#pragma unroll
    for (int p = 0; p < NPT; ++p) x[p] = *reinterpret_cast<const bf16x8*>(smem + (ks ? (pxaddr[TJ][p] ^ 64) : pxaddr[TJ][p]) + TI * PW * 128);
  };
  auto mma = [&](const bf16x8 (&w)[NCT], const bf16x8 (&x)[NPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int p = 0; p < NPT; ++p) acc[c][p] = DH_MFMA_16x16x32(w[c], x[p], acc[c][p]);
  };

  // Tile order (round 5): workgroup b runs on XCD b % 8.  With tile = b + round * G the 32 workgroups of an XCD held every eighth tile of a
  // 256-tile band, so no two neighbouring tiles shared an L2 and every halo row / column (10 x 34 patch for an 8 x 32 tile: 1.33 x) came
  // from the fabric again (rocprofv3 FETCH_SIZE: 1.25-1.35 x the map).  Each XCD now walks its own contiguous eighth of the tiles, 32 at
  // a time: horizontal neighbours load together, the row of tiles above was loaded one round earlier (1.4 MB ago in a 4 MiB L2).
  const bool xcd_walk = (G & 7) == 0;
  const int GS = xcd_walk ? (G >> 3) : G;
  const int per_xcd = (g.sp_items + 7) >> 3;
  const int xbase = xcd_walk ? (int)(blockIdx.x & 7u) * per_xcd : 0;
  const int xend = xcd_walk ? (xbase + per_xcd < g.sp_items ? xbase + per_xcd : g.sp_items) : g.sp_items;
  int sp = xcd_walk ? xbase + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (sp >= xend) return;
  issue_patch(sp, 0);
  if constexpr (!DGRAD) {
    if (tid < 64) reinterpret_cast<float*>(smem + RWBASE)[tid] = a.bias ? a.bias[tid] : 0.f;
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  c64_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (;;) {
    const int nsp = sp + GS;
    const bool has_next = nsp < xend;
    if constexpr (DGRAD) {
      issue_rw(sp);                                // this tile's mask / old value (its own reads of the region ended with its last epilogue)
      if (a.mask_bits) issue_bits(sp, buf);
      if constexpr (FUSE8) issue_img(sp, buf);     // this tile's image patch (its buffer's last reads ended two hand-off barriers ago)
    }
    if (has_next) issue_patch(nsp, buf ^ 1);       // the other buffer was released by the barrier that ended the previous tile
    // ---- 18 half-taps (tap, k-slice); pixel fragments software-pipelined TWO half-taps ahead through three rotating
    //      register sets (a ds_read_b128 under load takes longer than the 8 MFMAs of one half-tap)
    auto rd = [&](bf16x8 (&x)[NPT], auto hc) __attribute__((always_inline)) {
      constexpr int H = decltype(hc)::value;
      read_px(x, std::integral_constant<int, H / 2>{}, H & 1);
    };
    auto half = [&](auto hc, bf16x8 (&cur)[NPT], bf16x8 (&)[NPT]) __attribute__((always_inline)) {
      constexpr int H = decltype(hc)::value;
      mma(wr[H / 2][H & 1], cur);
      if constexpr (H + 3 < 18) rd(cur, std::integral_constant<int, H + 3>{});       // refill the set just consumed
    };
    rd(xa, std::integral_constant<int, 0>{});
    rd(xb, std::integral_constant<int, 1>{});
    rd(xc, std::integral_constant<int, 2>{});
#define C64_H(H, X) half(std::integral_constant<int, H>{}, X, X)
    C64_H(0, xa); C64_H(1, xb); C64_H(2, xc); C64_H(3, xa); C64_H(4, xb); C64_H(5, xc);
    C64_H(6, xa); C64_H(7, xb); C64_H(8, xc); C64_H(9, xa); C64_H(10, xb); C64_H(11, xc);
    C64_H(12, xa); C64_H(13, xb); C64_H(14, xc); C64_H(15, xa); C64_H(16, xb); C64_H(17, xc);
#undef C64_H
    // ---- hand-off: the next patch has landed (own pieces) and everybody is done reading this one
    c64_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // ---- epilogue (its stores drain under the next tile's MFMAs)
    {
      int n, y0, x0;
      sp_coords(sp, n, y0, x0);
      static_assert(NCT == 2, "epilogue packs the two channel tiles of a lane into one 16-byte vector");
      const int cbase = wn * 32 + fq * 8;                     // this lane's 8 consecutive output channels
      float bv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) bv[r] = DGRAD ? 0.f : reinterpret_cast<const float*>(smem + RWBASE)[cbase + r];
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      u32x4 pk[NPT];                                         // packed outputs of this lane (zero outside the image), for the fused pool
#pragma unroll
      for (int p = 0; p < NPT; ++p) pk[p] = u32x4{0u, 0u, 0u, 0u};
      // Data gradient: the staged mask, then the staged old value, are folded into the accumulators for ALL fragments before the first
      // store — the compiler orders an LDS read behind everything vmcnt counts (it cannot tell a DMA from a store), so a read issued
      // after a store would wait out a full write latency.  Two passes keep the live set at one 16-byte vector per fragment.
      if constexpr (DGRAD) {
        if (a.mask_bits) {                           // byte (wn*4 + fq) of pixel t = this lane's 8 channels
          unsigned w4[NPT];
#pragma unroll
          for (int p = 0; p < NPT; ++p) w4[p] = *reinterpret_cast<const unsigned*>(smem + RWBASE + buf * 2048 + (wm * 64 + p * 16 + frow) * 8 + wn * 4);
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            const int byte = (int)(w4[p] >> (8 * fq));
#pragma unroll
            for (int r = 0; r < 8; ++r) {            // value & (bit ? ~0 : 0): v_bfe_i32 + v_and_b32
              const float v = acc[r >> 2][p][r & 3];
              acc[r >> 2][p][r & 3] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe(byte, r, 1));
            }
          }
        }
        if (a.mask) {
          uint4 m[NPT];
#pragma unroll
          for (int p = 0; p < NPT; ++p) m[p] = *reinterpret_cast<const uint4*>(smem + rwaddr + p * 1024);
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            const bf16_t* mp = reinterpret_cast<const bf16_t*>(&m[p]);
#pragma unroll
            for (int r = 0; r < 8; ++r) if (!(bf2f(mp[r]) > 0.f)) acc[r >> 2][p][r & 3] = 0.f;
          }
        }
        if (a.accumulate) {
          uint4 o[NPT];
#pragma unroll
          for (int p = 0; p < NPT; ++p) o[p] = *reinterpret_cast<const uint4*>(smem + rwaddr + 32768 + p * 1024);
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            const bf16_t* op = reinterpret_cast<const bf16_t*>(&o[p]);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r >> 2][p][r & 3] += bf2f(op[r]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const int t = wm * 64 + p * 16 + frow;
        const int y = y0 + t / TW, x = x0 + t % TW;
        const bool ok = y < a.H && x < a.W;
        const size_t o0 = (size_t)((n * a.H + y) * a.W + x) * (size_t)a.ldy + cbase;      // (resid / pool_y: dense calls only, ldy = 64)
        if (ok) {
          float v[8] = {acc[0][p][0], acc[0][p][1], acc[0][p][2], acc[0][p][3], acc[1][p][0], acc[1][p][1], acc[1][p][2], acc[1][p][3]};
          if (!DGRAD) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += bv[r];
            if (a.relu) {
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] = dh_relu(v[r]);
            }
            if (a.resid) {
              const uint4 in = *reinterpret_cast<const uint4*>(a.resid + o0);
              const bf16_t* rp = reinterpret_cast<const bf16_t*>(&in);
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += bf2f(rp[r]);
            }
          } else {
            // (mask and old value were folded into acc above)
          }
          const u32x4 tt = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          if (a.y)            // (uniform; NULL = the pool-only inference call / the folded first-layer gradient: the map is never written)
            __builtin_nontemporal_store(tt, reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(a.y) + o0));   // streamed: re-read only after it left the L2
          pk[p] = tt;
        }
        if constexpr (FUSE8) {                       // the tile of dX the store would have written (zeros outside the image), [pixel][64 channels];
          // 16-byte chunk c of pixel t sits at position c ^ ((t & 7) ^ ((t >> 3) & 1) << 2): with the plain layout every pixel starts on bank 0
          // (128-byte pitch) - the eight lanes of a ds_write_b128 group conflicted 8 ways and the transposing reads 4 ways; swizzled, neither does
          const int fz = (t & 7) ^ (((t >> 3) & 1) << 2);
          *reinterpret_cast<u32x4*>(smem + DXBASE + t * 128 + (((cbase >> 3) ^ fz) << 4)) = pk[p];
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (FUSE8) {
        // ---- conv1_1's weight + bias gradient of this tile: dW[tap][c][co] += sum over pixels of X8[pixel @ tap][c] * dX[pixel][co]
        f32x4 facc[3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) facc[mt] = *reinterpret_cast<const f32x4*>(smem + faddr + mt * 1024);      // (own slots: no hazard with other waves)
        __builtin_amdgcn_s_waitcnt(0xC07F);           // (lgkmcnt(0): this wave's tile writes are in LDS, its accumulators in registers)
        __builtin_amdgcn_s_barrier();                 // ... and so are everybody else's tile writes
        const char* im = smem + IMGBASE + buf * IMGBYTES;
        // lane 4q+p of 16-lane group g supplies pixel 8g + q (+ 4h) of the row: tap slot p of M tile mt (image side), channels 4p..4p+3 of
        // the wave's N tile (dX side) — conv_wgrad_c8.hip's fragment addressing with a 32-pixel row
        int ln = lane;
        asm volatile("" : "+v"(ln));                  // (as above: nothing of this block may be hoisted into long-lived registers)
        const int gg = ln >> 4, qq = (ln & 15) >> 2, pp = ln & 3;
        const int nt = wave & 3, r0 = (wave >> 2) * 4;
        const int ypix = 8 * gg + qq;                 // + row * 32 + 4 h: the pixel of the tile this lane supplies; its chunk 2 nt + (pp >> 1), swizzled as written
        // all 32 transposing reads first (64 registers: the main loop's pixel fragments and accumulators are dead here), then the 12 MFMAs
        // behind counted waits — row by row the section paid four LDS latencies per tile
        s16x4 xh[4][2][3], yh[4][2];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = r0 + rr;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
              const int tap = mt * 4 + pp;
              const int ti = tap / 3, tj = tap - ti * 3;
              const char* ad = tap < 9 ? im + ((row + ti) * PW + 8 * gg + qq + tj + 4 * h) * 16 : smem + C1BASE + (tap == 9 ? 0 : 8);
              xh[rr][h][mt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)ad);
            }
            const int pix = row * 32 + ypix + 4 * h;
            const int fz = (pix & 7) ^ (((pix >> 3) & 1) << 2);
            yh[rr][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(smem + DXBASE + pix * 128 + (((2 * nt + (pp >> 1)) ^ fz) << 4) + (pp & 1) * 8));
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const bf16x8 yf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(yh[rr][0], yh[rr][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) {
            const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xh[rr][0][mt], xh[rr][1][mt], 0, 1, 2, 3, 4, 5, 6, 7));
            facc[mt] = DH_MFMA_16x16x32(xf, yf, facc[mt]);
          }
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) *reinterpret_cast<f32x4*>(smem + faddr + mt * 1024) = facc[mt];
      }
      if (!DGRAD && a.pool_y) {
        // fused 2x2 / stride-2 SAME max-pool (pool1, net/sfd_net.py:132): wave rows (2*wm, 2*wm+1); fragments p = 0,1 are the top
        // row (x 0-15, 16-31), p + 2 the row below; the horizontal neighbour is lane ^ 1; even lanes store.  ReLU outputs only.
        const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int t = wm * 64 + p * 16 + frow;
          const int y = y0 + t / TW, x = x0 + t % TW;
          u32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned v = pkmax_relu(pk[p][e], pk[p + 2][e]);
            m[e] = pkmax_relu(v, dh_lane_xor1(v));
          }
          if (y < a.H && x < a.W && (frow & 1) == 0)
            *reinterpret_cast<u32x4*>(a.pool_y + ((size_t)((n * Hp + (y >> 1)) * Wp + (x >> 1))) * 64 + cbase) = m;
          if (a.pool_arg_out) {                      // (uniform) 2-bit arg-max codes of the window for the pool's backward: 2 bytes per lane
            u32x4 tr;
#pragma unroll
            for (int e = 0; e < 4; ++e) tr[e] = dh_lane_xor1(pk[p][e]);
            const unsigned codes = dh_argmax2x2_codes16(pk[p], tr, pk[p + 2], m);
            if (y < a.H && x < a.W && (frow & 1) == 0)
              *reinterpret_cast<unsigned short*>(a.pool_arg_out + ((size_t)((n * Hp + (y >> 1)) * Wp + (x >> 1))) * 16 + cbase / 4) = (unsigned short)codes;
          }
        }
      }
    }
    if (!has_next) break;
    sp = nsp;
    buf ^= 1;
    const int dir = buf ? PBYTES : -PBYTES;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int p = 0; p < NPT; ++p) pxaddr[j][p] += dir;
  }
  if constexpr (FUSE8) {
    // ---- eight waves -> one gradient in LDS -> global atomics (conv_wgrad_c8.hip).  Lane holds facc[mt][r] = dW[tap slot mt*4 + g][channel r][co]
    // of the wave's N tile: co = 16 (wave & 3) + (lane & 15); waves w and w + 4 own the same tile for different rows
    f32x4 facc[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) facc[mt] = *reinterpret_cast<const f32x4*>(smem + faddr + mt * 1024);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                   // every wave is past its last reads of the patch buffers: reuse buffer 0
    float* red = reinterpret_cast<float*>(smem);    // [12 slots][4 channels][64 co]
    for (int i = tid; i < 12 * 4 * 64; i += 512) red[i] = 0.f;
    __syncthreads();
    const int gg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[((mt * 4 + gg) * 4 + r) * 64 + (wave & 3) * 16 + (lane & 15)], facc[mt][r]);
    __syncthreads();
    for (int i = tid; i < 9 * 4 * 64; i += 512) {
      const int co = i & 63, c = (i >> 6) & 3, slot = i >> 8;
      if (c < a.fuse_cin_real) atomicAdd(a.fuse_dw + (size_t)(slot * a.fuse_cin_real + c) * 64 + co, red[i]);
    }
    if (a.fuse_db && tid < 64) atomicAdd(a.fuse_db + tid, red[(9 * 4 + 0) * 64 + tid]);
  }
}

int c64_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

bool c64_eligible(const ConvArgs& a) {
  if (!(a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dstride == 1 && a.pad_t == 1 && a.pad_l == 1)) return false;
  if (a.H != a.Ho || a.W != a.Wo || a.C != 64 || a.Co != 64 || a.out_f32 || a.Kpad != 576) return false;
  if (a.relu && a.relu_co < a.Co) return false;                                    // partial ReLU: the streaming GEMM's epilogue only
  if (a.strided() && (a.resid || a.pool_y || a.mask_bits || a.bits_out)) return false;      // views: plain conv / masked, accumulating data gradient
  const int ldmax = a.ldx > a.ldy ? (a.ldx > a.ldm ? a.ldx : a.ldm) : (a.ldy > a.ldm ? a.ldy : a.ldm);
  if ((a.ldx | a.ldy | a.ldm) & 7) return false;
  const double util = (double)a.H * a.W / ((double)((a.H + 7) / 8 * 8) * (double)((a.W + 31) / 32 * 32));
  return util >= 0.78 && (int64_t)a.N * a.H * a.W * ldmax * 2 < (1ll << 32);
}

template <bool DGRAD, bool FUSE8 = false>
int launch_c64(const ConvArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * ((10 * 34 + 7) / 8) * 1024 + (FUSE8 ? 75776 : (DGRAD ? 65536 : 256));      // patches + (mask, old) tiles | bias; FUSE8: 160 KiB in all
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel<DGRAD, FUSE8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  C64Geom g{};
  g.tiles_x = (a.W + 31) / 32;
  g.tiles_y = (a.H + 7) / 8;
  g.sp_items = a.N * g.tiles_x * g.tiles_y;
  g.div_tx = make_fastdiv(g.tiles_x);
  g.div_txy = make_fastdiv(g.tiles_x * g.tiles_y);
  int G = c64_cu_count();
  if (g.sp_items < G) G = g.sp_items;
  hipLaunchKernelGGL((conv3x3_c64_kernel<DGRAD, FUSE8>), dim3(G), dim3(512), LDS, s, a, g);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

}  // namespace

int danhip_launch_conv_c64(const ConvArgs& a, hipStream_t s) {
  if (!c64_eligible(a)) return 1;
  const bool dgrad = !a.bias && !a.relu && !a.resid;
  if (!dgrad && (a.accumulate || a.mask || a.mask_bits)) return 1;
  if (a.mask && a.mask_bits) return 1;              // one mask form per call (the bits share the 16-bit mask's LDS region)
  if (a.fuse_dw) {                                  // conv1_1's weight gradient folded into this data gradient (bit-mask form, dense tensors)
    if (!dgrad || !a.mask_bits || a.accumulate || a.strided() || !a.fuse_x8 || a.fuse_cin_real < 1 || a.fuse_cin_real > 4) return 1;
    return launch_c64<true, true>(a, s);
  }
  return dgrad ? launch_c64<true>(a, s) : launch_c64<false>(a, s);
}

bool danhip_conv_c64_eligible(const ConvArgs& a) { return c64_eligible(a); }

const char* danhip_conv_c64_label(const ConvArgs& a, bool dgrad) {
  if (!c64_eligible(a)) return nullptr;
  return dgrad ? "conv3x3_c64_kernel<true>" : "conv3x3_c64_kernel<false>";
}
