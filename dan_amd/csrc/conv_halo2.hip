// Halo-reuse 3x3 / stride-1 convolution, second form: 512-pixel tiles and 32-channel chunks (round 3).
//
// Same job as conv_halo.hip (forward of every 3x3 'same' conv of the backbone / LFPN / context modules on maps >= 32 wide with
// Cout % 128 == 0, and — run on dY with the tap-flipped packing — their data gradients), different geometry.  tools/phase_probe.hip measured the
// two-wave-group skeleton stripped of everything else (MFMA-only = 100 %): what a phase costs is its LDS-DMA instructions (~33 MFMA-cycles
// each) and its fragment reads (~5 each), barriers are free.  conv_halo.hip runs 32 MFMAs against 16 reads and ~2.7 DMA pieces per wave and
// phase (probe: 75 %); a wave tile of 64 pixels x 128 channels needs 12 reads, and with the weight tile of a (tap, 32-channel) step being
// 8 KiB = ONE piece per wave, ~1.5 DMA pieces (probe: 82 %).  That tile is 16 x 32 pixels x 128 channels per workgroup; its halo patch only
// fits LDS twice (double buffer) when a chunk is 32 channels: [18 x 34 pixels][32 ch] = 39 KiB.
//
// LDS map: [patch 0][patch 1] (64-byte pixel rows), weight ring of 6 stages [128 co][32 ch] (64-byte rows), 8 KiB of bias (forward) /
// ReLU bits of the item (data gradient).  Both row kinds are 64 bytes, so a ds_read_b128 lane group (16 lanes) touches 16 rows x one of 4
// chunks: conflict-free with   weights: chunk ^= T[(row >> 3) & 3], T = {0, 3, 2, 1}   patch: chunk ^= ((hx >> 2) & 1) << 1   (hx = patch
// column; both found by exhaustive search over the b128 lane groups, applied on the DMA source side).  The patch key depends on the
// column only, so the nine taps of a chunk are immediate offsets from three per-lane addresses (one per tap column).
//
// Everything else follows conv_halo.hip: persistent 512-thread workgroup per CU, XCD-grouped item order, step sequence flattened across
// chunks and items, asm-free builtin LDS-DMA through buffer descriptors (out-of-range lane = zero fill = 'same' padding), counted vmcnt +
// raw s_barrier, two wave groups alternating a memory phase and an MFMA phase, weight fragment = MFMA A operand with channel tiles
// interleaved in pairs (a lane owns 8 consecutive output channels of a pixel: 16-byte NHWC stores), fused bias / ReLU / 2x2 max-pool /
// ReLU-bit-mask epilogue (forward), ReLU bits from LDS / 16-bit mask / accumulate epilogue (data gradient).
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace {

struct Halo2Geom {
  int tiles_x, tiles_y, sp_items, NB, cch, grouped;      // cch = C / 32
  int ablate;      // timing experiments (danhip_set_option("halo2_ablate")): 1 no patch DMA, 2 no weight DMA, 4 no output stores, 8 no fragment reads, 16 second barrier per step
  FastDiv div_tx, div_txy, div_nb;
#ifdef H2_TRACE
  unsigned* trace;   // tools/halo2_trace.hip: [2 groups][128 steps][4 stamps] shader-clock values of workgroup 0, waves 0 and 4
#endif
};

#ifdef H2_TRACE
// All-scalar stamp (the first form computed per-slot vector addresses and spilled ~90 registers): the clock and the LDS address are SGPRs; the two
// VGPRs live only inside the asm block.  The dynamic LDS segment starts at LDS address 0 (this file has no static __shared__).
#define H2_STAMP(slot)                                                                                              \
  do {                                                                                                              \
    if (blockIdx.x == 0 && (wave & 3) == 0 && step_idx < 128) {                                                      \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                                                   \
      const unsigned sa_ = (unsigned)(STRACE + (((wave >> 2) * 128 + step_idx) * 4 + (slot)) * 4);                  \
      unsigned va_, vd_;                                                                                            \
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_write_b32 %0, %1" : "=&v"(va_), "=&v"(vd_) : "s"(sa_), "s"(t_) : "memory"); \
    }                                                                                                               \
  } while (0)
// stamps inside the first item's epilogue: slots 0..7 per group behind the step stamps
#define H2_ESTAMP(k)                                                                                                \
  do {                                                                                                              \
    if (blockIdx.x == 0 && (wave & 3) == 0 && step_idx < 80) {                                                       \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                                                   \
      const unsigned sa_ = (unsigned)(STRACE + 4096 + ((wave >> 2) * 8 + (k)) * 4);                                 \
      unsigned va_, vd_;                                                                                            \
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_write_b32 %0, %1" : "=&v"(va_), "=&v"(vd_) : "s"(sa_), "s"(t_) : "memory"); \
    }                                                                                                               \
  } while (0)
#else
#define H2_STAMP(slot) do { } while (0)
#define H2_ESTAMP(k) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void h2_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
typedef __attribute__((ext_vector_type(4))) unsigned h2_u32x4;

__device__ __forceinline__ void h2_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// ---- epilogue arithmetic.  The first form of the epilogue was ~1500 VALU instructions and 32 ds_bpermute round trips per wave and item
// (tools/halo2_trace.hip: ~10,000 clocks per wave group and item, two groups back to back, no MFMA under them): fmaxf compiled to two
// v_max_f32 (canonicalise + max), the ReLU bits to ~7 instructions per value, their lane gather to LDS permutes.
// ReLU in one instruction; max(NaN, 0) = 0 like fmaxf.
__device__ __forceinline__ float h2_relu(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}
// Two packed 16-bit ReLU OUTPUTS (>= +0, never NaN: h2_relu) -> 1 per half that is > 0.  (min(x, 1) as unsigned 16-bit integers; clang
// lowers the generic vector min to compares and selects.)
typedef __attribute__((ext_vector_type(2))) unsigned short h2_u16x2;
typedef __attribute__((ext_vector_type(2))) unsigned h2_u32x2;
__device__ __forceinline__ h2_u16x2 h2_pos2(unsigned packed) {
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(packed), "s"(0x00010001u));
  return __builtin_bit_cast(h2_u16x2, r);
}
// bits |= (value r of the 8 packed ones > 0) << (SHIFT + r): one v_pk_min_u16 + one v_dot2_u32_u16 per register
template <int SHIFT>
__device__ __forceinline__ unsigned h2_pos_bits8_acc(const h2_u32x4& t, unsigned bits) {
  static_assert(SHIFT == 0 || SHIFT == 8, "weights are 16-bit");
#pragma unroll
  for (int e = 0; e < 4; ++e)
    bits = __builtin_amdgcn_udot2(h2_pos2(t[e]), h2_u16x2{(unsigned short)(1u << (SHIFT + 2 * e)), (unsigned short)(2u << (SHIFT + 2 * e))}, bits, false);
  return bits;
}
// 4 x 4 byte transpose over the four lanes that share a pixel (lane = fq * 16 + frow): lane fq holds byte q of channel pair q in `mine`
// (byte q = channels q * 32 + fq * 8 .. + 7) and returns bytes 4 fq .. 4 fq + 3 of the pixel's 16-byte row.  Two row swaps (VALU, gfx950)
// and two byte permutes instead of LDS permutes:
//   v_permlane16_swap(S, S): first = rows [0,0,2,2] (the even lane of each pair), second = rows [1,1,3,3] (the odd one)
//   v_permlane32_swap(X, X): first = the lower 32 lanes' X in both halves, second = the upper 32 lanes' X
__device__ __forceinline__ unsigned h2_transpose_bits(unsigned mine, int fq) {
  const h2_u32x2 s1 = __builtin_amdgcn_permlane16_swap(mine, mine, false, false);
  const unsigned k = (unsigned)(fq & 1) * 0x01010101u;
  const unsigned x = __builtin_amdgcn_perm(s1[1], s1[0], 0x06020400u + k);       // [E.b(k), O.b(k), E.b(k+2), O.b(k+2)]
  const h2_u32x2 s2 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  const unsigned j = (unsigned)(fq & 2) * 0x01010101u;
  return __builtin_amdgcn_perm(s2[1], s2[0], 0x05040100u + j);                    // [L.b(j), L.b(j+1), H.b(j), H.b(j+1)]
}

// The lane id, recomputed where it is needed: a value derived from threadIdx.x that is only used once per item was kept in a spilled register,
// and a scratch reload in the steady state costs a full vmcnt drain of the DMA queue.
__device__ __forceinline__ int h2_fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <bool DGRAD, bool POOL>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo2_kernel(const ConvArgs a, const Halo2Geom g) {
  constexpr int TH = 16, TW = 32, PW = TW + 2;
  constexpr int PROWS = (TH + 2) * PW;               // 612 patch pixels
  constexpr int PPIECES = (PROWS + 15) / 16;         // 39 DMA pieces of 16 pixel rows x 64 bytes
  constexpr int PBYTES = PPIECES * 1024;
  constexpr int PL = (PPIECES + 7) / 8;              // 5 patch pieces per wave
  constexpr int BN = 128;
  constexpr int WST = BN * 64;                       // one weight stage: [128 co][32 ch] = 8 KiB = one piece per wave
  constexpr int NSW = 6, D = NSW - 1;                // ring depth, prefetch distance in steps
  constexpr int NPT = 4, NCT = 8, NPAIR = 4;         // wave tile: 64 pixels (tile rows 2w, 2w+1) x 128 channels
  constexpr int WRING = 2 * PBYTES, SAUX = WRING + NSW * WST;
  constexpr int SPSRC = SAUX + 8192;                 // [PL + 1][512] patch source offsets of the item being fetched (patch_item_setup), weight lane offset
  [[maybe_unused]] constexpr int STRACE = SPSRC + (PL + 1) * 512 * 4;   // H2_TRACE builds only: 4 KiB of time stamps
  static_assert(STRACE + 4096 + 64 <= 160 * 1024, "LDS budget");
  static_assert(4 * PW * 64 + 2 * 64 + PBYTES < 65536, "ds_read immediate offsets");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;

  auto decode = [&](int v, int& sp, int& nb) __attribute__((always_inline)) -> bool {
    if (g.grouped) {
      const int r = v / G, b = v - r * G;
      const int xcd = b & 7, slot = b >> 3;
      const int spb = G / g.NB;
      const int q = (int)fdiv((unsigned)slot, g.div_nb);
      nb = slot - q * g.NB;
      sp = r * spb + xcd * ((G >> 3) / g.NB) + q;
    } else {
      sp = (int)fdiv((unsigned)v, g.div_nb);
      nb = v - sp * g.NB;
    }
    return sp < g.sp_items;
  };
  auto sp_coords = [&](int sp, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    n = (int)fdiv((unsigned)sp, g.div_txy);
    const int rem = sp - n * (g.tiles_x * g.tiles_y);
    const int ty = (int)fdiv((unsigned)rem, g.div_tx);
    y0 = ty * TH;
    x0 = (rem - ty * g.tiles_x) * TW;
  };

  // ---- patch DMA.  LDS row R = hy * PW + hx (64 bytes = 32 channels of pixel (y0-1+hy, x0-1+hx)); chunk c of the row sits at c ^ sx(hx).
  const int prow = lane >> 2, ppos = lane & 3;       // row inside a piece, chunk position
  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)a.C * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)((unsigned)a.Co * (unsigned)a.Kpad * 2u), 0x00020000);
  int p_v = blockIdx.x, p_cc = 0, p_idx = 0, p_sp, p_nb;
  bool p_ok = decode(p_v, p_sp, p_nb);
  // The five per-lane source offsets live in LDS, not in registers: the allocator spilled them around the epilogue and reloaded them from
  // scratch right before each DMA -- a scratch load is a VMEM op, so its s_waitcnt vmcnt(0) drained the whole prefetch queue three times
  // per chunk.  A ds_read_b32 waits on lgkmcnt only.
  // (Their LDS address is recomputed from the lane id at every use, and the weight DMA's lane offset lives in slot PL of the same table: both
  // were spilled too -- any long-lived per-lane value that is not an accumulator or a fragment address is a candidate.)
  auto psrc_slot = [&](int k) __attribute__((always_inline)) -> unsigned* {
    return reinterpret_cast<unsigned*>(smem + SPSRC) + k * 512 + wave * 64 + h2_fresh_lane();
  };
  auto patch_item_setup = [&]() __attribute__((always_inline)) {
    int n, y0, x0;
    sp_coords(p_sp, n, y0, x0);
    const int ln = h2_fresh_lane();                  // the piece geometry is recomputed per item, not hoisted into ~25 long-lived (spilled) registers
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      int piece = k * 8 + wave;
      if (piece > PPIECES - 1) piece = PPIECES - 1;  // duplicate of the last piece: uniform DMA count per wave
      const int row = piece * 16 + (ln >> 2);
      const int hy = row / PW, hx = row - hy * PW;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = row < PROWS && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const unsigned off = (unsigned)(((n * a.H + y) * a.W + x) * a.C) * 2u + (unsigned)(((ln & 3) ^ (((hx >> 2) & 1) << 1)) << 4);
      *psrc_slot(k) = ok ? off : 0xFFFFFFFFu;
    }
  };
  // One piece per step (taps 0 .. PL-1 of the chunk BEFORE the one that reads it): all five in one memory phase made that phase ~900 cycles
  // against a 512-cycle MFMA phase (ablation: the patch DMA cost 20 % of the kernel).  issue_patch_all() is the prologue's form.
  auto patch_advance = [&]() __attribute__((always_inline)) {
    ++p_idx;
    if (++p_cc == g.cch) {
      p_cc = 0;
      p_v += G;
      p_ok = decode(p_v, p_sp, p_nb);
      if (p_ok) patch_item_setup();
    }
  };
  auto issue_patch_piece = [&](auto kc) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    int piece = K * 8 + wave;
    if (piece > PPIECES - 1) piece = PPIECES - 1;
    if (!(g.ablate & 1)) h2_dma16(rsrc_x, *psrc_slot(K), (unsigned)(p_cc * 64), smem + (p_idx & 1) * PBYTES + piece * 1024);
    if (K == PL - 1) patch_advance();
  };
  auto issue_patch_all = [&]() __attribute__((always_inline)) {
    issue_patch_piece(std::integral_constant<int, 0>{}); issue_patch_piece(std::integral_constant<int, 1>{});
    issue_patch_piece(std::integral_constant<int, 2>{}); issue_patch_piece(std::integral_constant<int, 3>{});
    issue_patch_piece(std::integral_constant<int, 4>{});
    static_assert(PL == 5, "issue_patch_all issues PL pieces");
  };
  if (p_ok) patch_item_setup();

  // ---- weight DMA: stage (w_v, w_cc, w_tap): [128 co][32 ch] of tap w_tap; wave w moves rows 16 w .. 16 w + 15
  int w_v = blockIdx.x, w_cc = 0, w_tap = 0, w_idx = 0, w_sp, w_nb;
  bool w_ok = decode(w_v, w_sp, w_nb);
  {
    const int row = wave * 16 + prow;
    const int key = (4 - ((row >> 3) & 3)) & 3;       // T = {0, 3, 2, 1}
    *psrc_slot(PL) = (unsigned)(row * a.Kpad) * 2u + (unsigned)((ppos ^ key) << 4);
  }
  auto issue_w = [&]() __attribute__((always_inline)) {
    const unsigned soff = (unsigned)((w_nb * BN) * a.Kpad + w_tap * a.C + w_cc * 32) * 2u;
    if (!(g.ablate & 2)) h2_dma16(rsrc_w, *psrc_slot(PL), soff, smem + WRING + (w_idx % NSW) * WST + wave * 1024);
    ++w_idx;
    if (++w_tap == 9) {
      w_tap = 0;
      if (++w_cc == g.cch) {
        w_cc = 0;
        w_v += G;
        w_ok = decode(w_v, w_sp, w_nb);
      }
    }
  };

  // ---- ReLU bits of the data gradient's item: 512 pixels x 16 bytes = 8 pieces, one per wave, issued in step 1 of the item's first chunk
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.mask_bits ? a.mask_bits : reinterpret_cast<const unsigned char*>(a.x)), 0,
      (int)((unsigned)(a.N * a.H * a.W) * (unsigned)(a.Co / 8)), 0x00020000);
  auto issue_bits = [&](int sp, int nb) __attribute__((always_inline)) {
    int n, y0, x0;
    sp_coords(sp, n, y0, x0);
    const int t = wave * 64 + h2_fresh_lane();
    const int y = y0 + t / TW, x = x0 + t % TW;
    const unsigned off = (y < a.H && x < a.W) ? (unsigned)((n * a.H + y) * a.W + x) * (unsigned)(a.Co / 8) + (unsigned)(nb * (BN / 8)) : 0xFFFFFFFFu;
    h2_dma16(rsrc_b, off, 0u, smem + SAUX + wave * 1024);
  };

  // ---- fragment addresses
  const int frow = lane & 15, fq = lane >> 4;
  const int wrow0 = (frow >> 2) * 8 + (frow & 3);
  const int offW = WRING + wrow0 * 64 + ((fq ^ ((4 - ((wrow0 >> 3) & 3)) & 3)) << 4);      // + stage*WST + (c>>1)*2048 + (c&1)*256
  [[maybe_unused]] int xoff[3];                                       // tap column j: patch buffer 0, this wave's first tile row, pixel-tile half 0
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int hx = j + frow;                         // (+16 for the second half: the key's bit 2 of hx is unchanged)
    xoff[j] = ((2 * wave) * PW + hx) * 64 + ((fq ^ (((hx >> 2) & 1) << 1)) << 4);
  }

  // (Starting an item's accumulators from the MFMA's C operand -- bias or 0 -- instead of zeroing them in the epilogue was tried: the two
  // forms of the first step make a 128-register phi that the allocator answers with ~300 spilled registers.)
  f32x4 acc[NCT][NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int p = 0; p < NPT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 wf[NCT], xf[NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) wf[c] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < NPT; ++p) xf[p] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  auto load_frags = [&](auto tapc, int wbase, int pofs) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr int TI = TAP / 3, TJ = TAP % 3;
    if (g.ablate & 8) return;
    int xo;
    if constexpr (DGRAD) {                           // recomputed (9 VALU ops): the data gradient's build kept one xoff[] in scratch, see h2_fresh_lane
      const int l = h2_fresh_lane();
      const int hx = TJ + (l & 15);
      xo = ((2 * wave) * PW + hx) * 64 + (((l >> 4) ^ (((hx >> 2) & 1) << 1)) << 4);
    } else {
      xo = xoff[TJ];
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c) wf[c] = *reinterpret_cast<const bf16x8*>(smem + wbase + (c >> 1) * 2048 + (c & 1) * 256);
#pragma unroll
    for (int p = 0; p < NPT; ++p)                    // p = (row r = p >> 1, half h = p & 1)
      xf[p] = *reinterpret_cast<const bf16x8*>(smem + xo + pofs + (((p >> 1) + TI) * PW + (p & 1) * 16) * 64);
  };
  auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int p = 0; p < NPT; ++p) acc[c][p] = DH_MFMA_16x16x32(wf[c], xf[p], acc[c][p]);
  };
  // ---- prologue
  int c_v = blockIdx.x, c_sp, c_nb;
  bool c_ok = decode(c_v, c_sp, c_nb);
  if (!c_ok) return;
  issue_patch_all();
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (w_ok) issue_w();
  if constexpr (!DGRAD) {
    float* sb = reinterpret_cast<float*>(smem + SAUX);
    for (int i = tid; i < a.Co; i += 512) sb[i] = a.bias ? a.bias[i] : 0.f;
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  h2_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // Stores go through buffer descriptors: a 32-bit byte offset per pixel fragment (one v_add per store instead of 64-bit address arithmetic and
  // an exec-mask branch), a pixel outside the map has offset 2^31 = out of range = dropped (halo2_eligible: every output is <= 2^31 bytes;
  // 0xFFFFFFFF would wrap back into range with the instruction's immediate offset).
  constexpr unsigned OOB = 0x80000000u;
  int step_idx = 0;                                  // running step number (weight ring stage = step_idx % NSW)
  auto epilogue = [&]() __attribute__((always_inline)) {
    H2_ESTAMP(0);
    int n, y0, x0;
    sp_coords(c_sp, n, y0, x0);
    const int el = h2_fresh_lane();                  // recomputed: keeps the per-pixel geometry below out of long-lived (spilled) registers
    const int frow = el & 15, fq = el >> 4;
    const int cb = c_nb * BN + fq * 8;               // this lane's 8 consecutive channels of pair 0 (+32 per pair)
    const __amdgpu_buffer_rsrc_t rsrc_y =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)a.Co * 2u), 0x00020000);
    bool okp[NPT];
    unsigned pix[NPT];                               // pixel index n * H * W + y * W + x
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const int t = wave * 64 + p * 16 + frow;
      const int y = y0 + t / TW, x = x0 + t % TW;
      okp[p] = y < a.H && x < a.W;
      pix[p] = (unsigned)((n * a.H + y) * a.W + x);
    }
    if constexpr (DGRAD) {
      if (a.mask_bits) {                             // the staged bit mask: 16 bytes per pixel = the workgroup's 128 channels
        const int sh = 8 * fq;                       // byte q * 4 + fq = channels cb + q * 32 .. + 7
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          const int t = wave * 64 + p * 16 + frow;
          const uint4 bb = *reinterpret_cast<const uint4*>(smem + SAUX + t * 16);
          const unsigned wq[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
          for (int q = 0; q < NPAIR; ++q) {
            const int byte = (int)(wq[q] >> sh);
#pragma unroll
            for (int r = 0; r < 8; ++r) {            // value & (bit ? ~0 : 0): v_bfe_i32 + v_and_b32
              const float v = acc[2 * q + (r >> 2)][p][r & 3];
              acc[2 * q + (r >> 2)][p][r & 3] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe(byte, r, 1));
            }
          }
        }
      }
      // 16-bit mask, then old value: read-modify inputs from HBM, two pixel fragments (8 x 16-byte loads) in flight per kind; one kind at a
      // time, so that one 32-register array is live beside the 128 accumulators (both at once spilled ~90 VGPRs)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if (a.mask) {
          uint4 in0[2][NPAIR];
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < NPAIR; ++q)
              if (okp[ph * 2 + pp]) in0[pp][q] = *reinterpret_cast<const uint4*>(a.mask + (size_t)pix[ph * 2 + pp] * a.Co + cb + q * 32);
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
              if (!okp[ph * 2 + pp]) continue;
              const bf16_t* mp = reinterpret_cast<const bf16_t*>(&in0[pp][q]);
#pragma unroll
              for (int r = 0; r < 8; ++r) if (!(bf2f(mp[r]) > 0.f)) acc[2 * q + (r >> 2)][ph * 2 + pp][r & 3] = 0.f;
            }
        }
        uint4 in1[2][NPAIR];
        if (a.accumulate) {
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < NPAIR; ++q)
              if (okp[ph * 2 + pp])
                in1[pp][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.y) + (size_t)pix[ph * 2 + pp] * a.Co + cb + q * 32);
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int p = ph * 2 + pp;
          const unsigned yo = okp[p] && !(g.ablate & 4) ? (pix[p] * (unsigned)a.Co + (unsigned)cb) * 2u : OOB;
#pragma unroll
          for (int q = 0; q < NPAIR; ++q) {
            float v[8] = {acc[2 * q][p][0], acc[2 * q][p][1], acc[2 * q][p][2], acc[2 * q][p][3],
                          acc[2 * q + 1][p][0], acc[2 * q + 1][p][1], acc[2 * q + 1][p][2], acc[2 * q + 1][p][3]};
            if (a.accumulate && okp[p]) {
              const bf16_t* op = reinterpret_cast<const bf16_t*>(&in1[pp][q]);
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += bf2f(op[r]);
            }
            const h2_u32x4 tt = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            __builtin_amdgcn_raw_buffer_store_b128(tt, rsrc_y, (int)(yo + q * 64), 0, 0);
            acc[2 * q][p] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[2 * q + 1][p] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else {
      // forward: channel pairs outermost, so that only one pair's packed outputs (for the fused pool) and bias values are live at a time
      const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
      unsigned yo[NPT];
#pragma unroll
      for (int p = 0; p < NPT; ++p) yo[p] = okp[p] && !(g.ablate & 4) ? (pix[p] * (unsigned)a.Co + (unsigned)cb) * 2u : OOB;
      [[maybe_unused]] unsigned ppix[2];             // fused pool: pooled pixel index of fragments 0, 1 (even lanes)
      [[maybe_unused]] bool pok[2];
      [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrc_p;
      if constexpr (POOL) {
        rsrc_p = __builtin_amdgcn_make_buffer_rsrc(a.pool_y, 0, (int)((unsigned)(a.N * Hp * Wp) * (unsigned)a.Co * 2u), 0x00020000);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int t = wave * 64 + p * 16 + frow;
          const int y = y0 + t / TW, x = x0 + t % TW;
          pok[p] = okp[p] && (frow & 1) == 0;
          ppix[p] = (unsigned)((n * Hp + (y >> 1)) * Wp + (x >> 1));
        }
      }
      unsigned pbA[NPT] = {0u, 0u, 0u, 0u}, pbB[NPT] = {0u, 0u, 0u, 0u};   // ReLU-bit bytes of a pixel fragment: pairs 0, 1 (bits 0..15) and 2, 3
      [[maybe_unused]] unsigned pb2A[2] = {0u, 0u}, pb2B[2] = {0u, 0u};
      H2_ESTAMP(1);
#pragma unroll
      for (int q = 0; q < NPAIR; ++q) {
        H2_ESTAMP(2 + q);
        float bv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[r] = reinterpret_cast<const float*>(smem + SAUX)[cb + q * 32 + r];
        [[maybe_unused]] h2_u32x4 pkq[NPT];
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          float v[8] = {acc[2 * q][p][0], acc[2 * q][p][1], acc[2 * q][p][2], acc[2 * q][p][3],
                        acc[2 * q + 1][p][0], acc[2 * q + 1][p][1], acc[2 * q + 1][p][2], acc[2 * q + 1][p][3]};
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += bv[r];
          acc[2 * q][p] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[2 * q + 1][p] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (a.relu) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = h2_relu(v[r]);
          }
          h2_u32x4 tt = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          __builtin_amdgcn_raw_buffer_store_b128(tt, rsrc_y, (int)(yo[p] + q * 64), 0, 0);
          if constexpr (POOL) {
            if (!okp[p]) tt = h2_u32x4{0u, 0u, 0u, 0u};
            pkq[p] = tt;
          }
          if (a.bits_out) {                          // (uniform; only with a.relu: halo2_eligible) bits of pixels outside the map are never stored
            if (q < 2) pbA[p] = q == 0 ? h2_pos_bits8_acc<0>(tt, pbA[p]) : h2_pos_bits8_acc<8>(tt, pbA[p]);
            else pbB[p] = q == 2 ? h2_pos_bits8_acc<0>(tt, pbB[p]) : h2_pos_bits8_acc<8>(tt, pbB[p]);
          }
        }
        if constexpr (POOL) {
          // 2x2 / stride-2 SAME max-pool of the wave's two rows from the packed ReLU outputs: fragment p (row 0) against p + 2 (row 1),
          // horizontal neighbour in lane ^ 1; even lanes store
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            h2_u32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned v = pkmax_relu(pkq[p][e], pkq[p + 2][e]);
              m[e] = pkmax_relu(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]: lane ^ 1
            }
            const unsigned po = pok[p] ? (ppix[p] * (unsigned)a.Co + (unsigned)cb) * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(m, rsrc_p, (int)(po + q * 64), 0, 0);
            if (a.pool_bits_out) {
              if (q < 2) pb2A[p] = q == 0 ? h2_pos_bits8_acc<0>(m, pb2A[p]) : h2_pos_bits8_acc<8>(m, pb2A[p]);
              else pb2B[p] = q == 2 ? h2_pos_bits8_acc<0>(m, pb2B[p]) : h2_pos_bits8_acc<8>(m, pb2B[p]);
            }
          }
        }
      }
      H2_ESTAMP(6);
      if (a.bits_out) {                              // (uniform) ReLU bit mask of y: lane fq stores bytes 4 fq .. 4 fq + 3 of its pixel's 16
        const __amdgpu_buffer_rsrc_t rsrc_b =
            __builtin_amdgcn_make_buffer_rsrc(a.bits_out, 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)(a.Co / 8)), 0x00020000);
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          const unsigned w = h2_transpose_bits(pbA[p] | (pbB[p] << 16), fq);
          const unsigned bo = okp[p] ? pix[p] * (unsigned)(a.Co / 8) + (unsigned)(c_nb * (BN / 8) + fq * 4) : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(w, rsrc_b, (int)bo, 0, 0);
        }
      }
      if constexpr (POOL) {
        if (a.pool_bits_out) {
          const __amdgpu_buffer_rsrc_t rsrc_b =
              __builtin_amdgcn_make_buffer_rsrc(a.pool_bits_out, 0, (int)((unsigned)(a.N * Hp * Wp) * (unsigned)(a.Co / 8)), 0x00020000);
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const unsigned w = h2_transpose_bits(pb2A[p] | (pb2B[p] << 16), fq);
            const unsigned bo = pok[p] ? ppix[p] * (unsigned)(a.Co / 8) + (unsigned)(c_nb * (BN / 8) + fq * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(w, rsrc_b, (int)bo, 0, 0);
          }
        }
      }
    }
    H2_ESTAMP(7);
    c_v += G;
    c_ok = decode(c_v, c_sp, c_nb);
  };

  // ---- main loop: cycle c = step c of this block's flattened (item, chunk, tap) sequence.  Per cycle every wave issues, in this order,
  // [patch piece (taps 0..4) | ReLU-bit piece (data gradient, tap 1 of an item's first chunk)] then its piece of weight stage c + D.
  // Hand-off rule (conv_halo.hip): before b1 of cycle c every wave has waited for its piece of W(c+1) — and, the extra pieces being
  // OLDER than the weight piece of their cycle, for every extra piece issued up to cycle c + 1 - D.  Operations younger than W(c+1):
  //   group A (has issued through cycle c):       W(c+2 .. c+D) = D - 1, + the extra pieces of cycles c + 2 - D .. c      (D - 1 cycles)
  //   group B (has issued through cycle c - 1):   W(c+2 .. c+D-1) = D - 2, + the extra pieces of cycles c + 2 - D .. c - 1 (D - 2 cycles)
  // `hist` keeps the number of extra pieces per cycle, newest in the low nibble.  (The epilogue's stores are not counted: the trace of
  // tools/halo2_trace.hip shows no wait on their acknowledgements in the steps after an epilogue.)  A patch piece of tap 4 is therefore complete before b1
  // of tap 8 — the barrier after which group B reads tap 0 of the next chunk.
  int chunk = 0, cc = 0;
  unsigned hist = 0;
  auto extra_in = [&](int cycles) __attribute__((always_inline)) -> int {      // sum of the newest `cycles` nibbles
    int n = 0;
#pragma unroll
    for (int i = 0; i < cycles; ++i) n += (int)((hist >> (4 * i)) & 15u);
    return n;
  };
  auto wait_young = [&](int n) __attribute__((always_inline)) {                  // n (wave-uniform) operations may stay in flight
    switch (n) {
      case 0: h2_wait_vmcnt<0>(); break;
      case 1: h2_wait_vmcnt<1>(); break;
      case 2: h2_wait_vmcnt<2>(); break;
      case 3: h2_wait_vmcnt<3>(); break;
      case 4: h2_wait_vmcnt<4>(); break;
      case 5: h2_wait_vmcnt<5>(); break;
      case 6: h2_wait_vmcnt<6>(); break;
      case 7: h2_wait_vmcnt<7>(); break;
      case 8: h2_wait_vmcnt<8>(); break;
      case 9: h2_wait_vmcnt<9>(); break;
      default: h2_wait_vmcnt<10>(); break;
    }
  };
  auto issue_extras = [&](auto tapc, bool bits_ok) __attribute__((always_inline)) -> int {
    constexpr int TAP = decltype(tapc)::value;
    int n = 0;
    if constexpr (TAP < PL) {
      if (p_ok) { issue_patch_piece(std::integral_constant<int, TAP>{}); n += 1; }
    }
    if constexpr (DGRAD && TAP == 1) {
      if (bits_ok) { issue_bits(c_sp, c_nb); n += 1; }
    }
    return n;
  };

  if (wave < 4) {
    for (;;) {
      const int pofs = (chunk & 1) * PBYTES;
      auto cycle = [&](auto tapc) __attribute__((always_inline)) {
        const int wbase = offW + (step_idx % NSW) * WST;
        H2_STAMP(0);
        load_frags(tapc, wbase, pofs);
        __builtin_amdgcn_sched_barrier(0);
        const int n_now = issue_extras(tapc, cc == 0 && a.mask_bits != nullptr);
        hist = (hist << 4) | (unsigned)n_now;
        const bool more_w = w_ok;
        if (more_w) issue_w();
        if (!more_w) h2_wait_vmcnt<0>();
        else wait_young(D - 1 + extra_in(D - 1));
        __builtin_amdgcn_s_waitcnt(0xC07F);
        H2_STAMP(1);
        __builtin_amdgcn_s_barrier();                // b1
        H2_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        mma();
        __builtin_amdgcn_sched_barrier(0);
        H2_STAMP(3);
        // The item's epilogue right after its last MFMAs, BEFORE b2: group B runs its own in the memory phase that is in progress, so the two
        // (~1000 instructions each, the only stretch without MFMAs) overlap instead of following each other in two steps.
        if (decltype(tapc)::value == 8 && cc + 1 == g.cch) epilogue();
        if (g.ablate & 16) __builtin_amdgcn_s_barrier();         // b2: timing experiment only (see conv_halo.hip's hand-off notes)
        ++step_idx;
      };
      cycle(std::integral_constant<int, 0>{}); cycle(std::integral_constant<int, 1>{}); cycle(std::integral_constant<int, 2>{});
      cycle(std::integral_constant<int, 3>{}); cycle(std::integral_constant<int, 4>{}); cycle(std::integral_constant<int, 5>{});
      cycle(std::integral_constant<int, 6>{}); cycle(std::integral_constant<int, 7>{}); cycle(std::integral_constant<int, 8>{});
      ++chunk;
      if (++cc == g.cch) {
        cc = 0;
        if (!c_ok) break;                            // (the epilogue advanced c_v)
      }
    }
  } else {
    bool w_prev = true;
    load_frags(std::integral_constant<int, 0>{}, offW, 0);
    for (;;) {
      const int pofs = (chunk & 1) * PBYTES;
      bool last = false;
      auto cycle = [&](auto tapc) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tapc)::value;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        H2_STAMP(0);
        __builtin_amdgcn_sched_barrier(0);
        mma();
        __builtin_amdgcn_sched_barrier(0);
        H2_STAMP(1);
        if (!w_prev) h2_wait_vmcnt<0>();
        else wait_young(D - 2 + extra_in(D - 2));
        __builtin_amdgcn_s_barrier();                // b1
        H2_STAMP(2);
        constexpr int NTAP = (TAP + 1) % 9;
        const int npofs = TAP == 8 ? (PBYTES - pofs) : pofs;
        const int nwbase = offW + ((step_idx + 1) % NSW) * WST;
        if (TAP == 8 && cc + 1 == g.cch) {           // the item ended with this step
          epilogue();
          if (!c_ok) last = true;
        }
        __builtin_amdgcn_sched_barrier(0);
        load_frags(std::integral_constant<int, NTAP>{}, nwbase, npofs);
        __builtin_amdgcn_sched_barrier(0);
        const int n_now = issue_extras(tapc, cc == 0 && a.mask_bits != nullptr && !last);
        hist = (hist << 4) | (unsigned)n_now;
        const bool more_w = w_ok;
        if (more_w) issue_w();
        w_prev = more_w;
        H2_STAMP(3);
        if (g.ablate & 16) __builtin_amdgcn_s_barrier();         // b2: timing experiment only (see conv_halo.hip's hand-off notes)
        ++step_idx;
      };
      cycle(std::integral_constant<int, 0>{}); cycle(std::integral_constant<int, 1>{}); cycle(std::integral_constant<int, 2>{});
      cycle(std::integral_constant<int, 3>{}); cycle(std::integral_constant<int, 4>{}); cycle(std::integral_constant<int, 5>{});
      cycle(std::integral_constant<int, 6>{}); cycle(std::integral_constant<int, 7>{}); cycle(std::integral_constant<int, 8>{});
      ++chunk;
      if (++cc == g.cch) cc = 0;
      if (last) break;
    }
  }
  h2_wait_vmcnt<0>();
#ifdef H2_TRACE
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = tid; i < 1024 + 16; i += 512) g.trace[i] = reinterpret_cast<const unsigned*>(smem + STRACE)[i];
#endif
}

#ifdef H2_TRACE
unsigned* h2_trace_buffer() {
  static unsigned* p = [] { void* q = nullptr; (void)hipMalloc(&q, 4096 + 64); (void)hipMemset(q, 0, 4096 + 64); return (unsigned*)q; }();
  return p;
}
#endif

int h2_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

bool halo2_eligible(const ConvArgs& a) {
  if (!danhip_option("halo2")) return false;        // off by default (danhip_set_option("halo2", 1) / DANHIP_HALO2=1): see DESIGN.md section 7
  if (!(a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dstride == 1 && a.pad_t == 1 && a.pad_l == 1)) return false;
  if (a.H != a.Ho || a.W != a.Wo) return false;
  if (a.C % 32 != 0 || a.C < 64 || a.Co % 128 != 0 || a.Co > 2048 || a.Kpad != 9 * a.C) return false;
  if (a.out_f32 || a.resid) return false;
  if ((a.bits_out || a.pool_bits_out) && !a.relu) return false;      // the emitted bits are those of ReLU outputs (h2_pos2)
  if ((int64_t)a.N * a.H * a.W * a.Co * 2 > (1ll << 31)) return false;  // 32-bit store offsets with 2^31 as the out-of-range value
  if ((int64_t)a.Co * a.Kpad >= (1ll << 31)) return false;
  // 16 x 32 tiles: the map must fill them (640 / 320 / 160 wide maps do; 80 x 80 keeps the 16 x 16 tiles of conv_halo.hip)
  const double ph = (double)((a.H + 15) / 16 * 16), pw = (double)((a.W + 31) / 32 * 32);
  if ((double)a.H * a.W / (ph * pw) < 0.9) return false;
  const long items = (long)a.N * ((a.H + 15) / 16) * ((a.W + 31) / 32) * (a.Co / 128);
  return items >= 2 * h2_cu_count();                 // fewer items than two rounds: the 256-pixel tiles quantise better
}

template <bool DGRAD, bool POOL>
int launch_halo2(const ConvArgs& a, hipStream_t s) {
#ifdef H2_TRACE
  constexpr int LDS = 2 * 39 * 1024 + 6 * 8192 + 8192 + 6 * 512 * 4 + 4096 + 64;
#else
  constexpr int LDS = 2 * 39 * 1024 + 6 * 8192 + 8192 + 6 * 512 * 4;
#endif
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo2_kernel<DGRAD, POOL>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  Halo2Geom g{};
  g.tiles_x = (a.W + 31) / 32;
  g.tiles_y = (a.H + 15) / 16;
  g.sp_items = a.N * g.tiles_x * g.tiles_y;
  g.NB = a.Co / 128;
  g.cch = a.C / 32;
  g.ablate = danhip_option("halo2_ablate");
#ifdef H2_TRACE
  g.trace = h2_trace_buffer();
#endif
  g.div_tx = make_fastdiv(g.tiles_x);
  g.div_txy = make_fastdiv(g.tiles_x * g.tiles_y);
  g.div_nb = make_fastdiv(g.NB);
  const long items = (long)g.sp_items * g.NB;
  int G = h2_cu_count();
  g.grouped = (items >= G && (G % 8) == 0 && ((G / 8) % g.NB) == 0 && g.NB > 1) ? 1 : 0;
  if (items < G) G = (int)items;
  hipLaunchKernelGGL((conv3x3_halo2_kernel<DGRAD, POOL>), dim3(G), dim3(512), LDS, s, a, g);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

}  // namespace

bool danhip_conv_halo2_eligible(const ConvArgs& a) { return halo2_eligible(a); }

const char* danhip_conv_halo2_label(const ConvArgs& a, bool dgrad) {
  if (!halo2_eligible(a)) return nullptr;
  if (dgrad) return "conv3x3_halo2_kernel<true, false>";
  return a.pool_y ? "conv3x3_halo2_kernel<false, true>" : "conv3x3_halo2_kernel<false, false>";
}

// DANHIP_OK when launched, 1 when the shape is not eligible (caller falls back to conv_halo.hip), negative on a launch error.
int danhip_launch_conv_halo2(const ConvArgs& a, hipStream_t s) {
  if (!halo2_eligible(a)) return 1;
  const bool dgrad = !a.bias && !a.relu;
  if (!dgrad && (a.accumulate || a.mask || a.mask_bits)) return 1;
  if (dgrad && (a.pool_y || a.bits_out)) return 1;
  if (a.mask && a.mask_bits) return 1;
  if (dgrad) return launch_halo2<true, false>(a, s);
  if (a.pool_y) {
    if (!(a.bias && a.relu)) return 1;
    return launch_halo2<false, true>(a, s);
  }
  return launch_halo2<false, false>(a, s);
}
