// Halo-reuse 3x3 / stride-1 convolution on MFMA for gfx950 (bf16 NHWC in, fp32 accumulate) — the kernel that carries the
// backbone (net/sfd_net.py:127-156 conv1_2 .. conv4_3), LFPN fused convs (net/pb_net.py:185-226, net/danet.py:339-380),
// shared head convs (net/danet.py:469-532), the PyramidBox CPM (net/pb_net.py:158-183), the thin loc/cls heads
// (net/sfd_net.py:159-219) and, run on dY with tap-flipped weights, their data gradients.
//
// One PERSISTENT 512-thread workgroup per CU walks work items (spatial tile TH x TW of one image) x (BN output channels).
// For every 64-channel chunk of the input the (TH+2) x (TW+2) halo patch is DMA'd into LDS ONCE ([pixel][64 ch] = 128-byte
// rows) and all nine taps read it at shifted offsets: the activation is fetched once instead of nine times (the flat-M
// kernel conv_igemm.hip re-gathers it per tap).  Weight tiles [TPS taps][BN][64] stream through an NSW-deep LDS ring
// (TPS = taps per step: 1 for the 128-wide tile, 3 = one kernel row for the 64-wide tile so a step still carries 48 MFMAs).
// All DMAs are buffer_load ... lds (16 bytes per lane); padding pixels are lanes whose offset is out of the descriptor's
// range (hardware zero fill).  The loop never drains the DMA queue: counted s_waitcnt vmcnt(N) + raw s_barrier.
// The step sequence is flattened across chunks and items, so the next item's first patch and weights are already landing
// while the current item's epilogue runs.
//
// Two wave groups (waves 0-3 / 4-7, one wave of each per SIMD) alternate phases:
//     A:  mem(c)   | b1 | MFMA(c) | b2          B:  MFMA(c) | b1 | mem(c+1) | b2
// The MFMA phase is nothing but the step's MFMAs; the mem phase reads the next fragments from LDS, issues the DMAs of
// step c+D and runs the item epilogue — DMA issue, LDS latency and barrier skew of one group hide under the other's MFMAs.
//
// MFMA: v_mfma_f32_16x16x32_bf16, weight fragment = A operand, pixel fragment = B operand -> a lane owns 4 consecutive
// output channels of one pixel (8-byte bf16 / 16-byte fp32 stores into NHWC).
// LDS swizzles (16-byte chunks, applied on the DMA source side and on the read side): weight rows by (row & 7), patch rows
// by the patch COLUMN (hx & 7) so that a tap's row shift is a pure immediate offset; both conflict-free for ds_read_b128.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace {

struct HaloGeom {
  int tiles_x, tiles_y;     // spatial tiles per image
  int sp_items;             // N * tiles_y * tiles_x
  int NB;                   // output-channel blocks (Co / BN)
  int cch;                  // 64-channel chunks of the input (C / 64)
  int grouped;              // 1: XCD-grouped item mapping (the NB blocks of one spatial tile run on one XCD)
  int crem;                 // C % 64 != 0: 16-byte pieces of the LAST input chunk that exist ((C % 64) / 8); 0 = whole chunks only
  int b2;                   // 1: second barrier per step (option "halo_b2"); see the main loop's hand-off notes
  int fast;                 // 1: the lean epilogue (16-bit output of <= 2^31 bytes, no 16-bit ReLU mask, no residual)
  FastDiv div_tx, div_txy, div_nb;
#ifdef H_TRACE
  unsigned* trace;          // tools/halo2_trace.hip -DTRACE_HALO1: [2 groups][128 steps][4 stamps] shader clocks of workgroup 0, waves 0 and 4
#endif
};

#ifdef H_TRACE
// All-scalar time stamp into the upper half of the bias / bit staging area (forward with Co <= 1024 and the bit-mask data gradient of
// 256-pixel tiles use only its lower 4 KiB); the dynamic LDS segment starts at LDS address 0.
#define H_STAMP(slot)                                                                                               \
  do {                                                                                                              \
    if (blockIdx.x == 0 && (wave & 3) == 0 && tr_step < 128) {                                                       \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                                                   \
      const unsigned sa_ = (unsigned)(SBIAS + 4096 + (((wave >> 2) * 128 + tr_step) * 4 + (slot)) * 4);             \
      unsigned va_, vd_;                                                                                            \
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_write_b32 %0, %1" : "=&v"(va_), "=&v"(vd_) : "s"(sa_), "s"(t_) : "memory"); \
    }                                                                                                               \
  } while (0)
#define H_STEP_DONE() (++tr_step)
#else
#define H_STAMP(slot) do { } while (0)
#define H_STEP_DONE() do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue of one lane's 4 consecutive channels (Co % 64 == 0: always a full, aligned quad).
//   forward : v = acc + bias; relu?; (+ residual, bf16) -> bf16 or fp32
//   dgrad   : v = acc; * (mask > 0)?; (+= old)?          -> bf16
template <bool DGRAD>
__device__ __forceinline__ void halo_finish4(const ConvArgs& a, const f32x4& acc, const float4& b, const uint2& in0, const uint2& in1, size_t o) {
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (!DGRAD) {
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    if (a.relu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dh_relu(v[r]);
    }
    if (a.out_f32) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + o) = make_float4(v[0], v[1], v[2], v[3]);
      return;
    }
    if (a.resid) {
      const bf16_t* rp = reinterpret_cast<const bf16_t*>(&in0);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bf2f(rp[r]);
    }
  } else {
    if (a.mask) {
      const bf16_t* mp = reinterpret_cast<const bf16_t*>(&in0);
#pragma unroll
      for (int r = 0; r < 4; ++r) if (!(bf2f(mp[r]) > 0.f)) v[r] = 0.f;
    }
    if (a.accumulate) {
      const bf16_t* op = reinterpret_cast<const bf16_t*>(&in1);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bf2f(op[r]);
    }
  }
  uint2 t;
  t.x = pack2bf(v[0], v[1]);
  t.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.y) + o) = t;
}

// 8 consecutive channels of one pixel (two interleaved channel tiles): 16-byte mask / residual / old-value reads, one
// 16-byte store (fp32 output: two float4 stores).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <bool DGRAD>
__device__ __forceinline__ u32x4 halo_finish8(const ConvArgs& a, const f32x4& lo, const f32x4& hi, const float (&b)[8], const uint4& in0,
                                              const uint4& in1, size_t o) {
  float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  if (!DGRAD) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += b[r];
    if (a.relu) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = dh_relu(v[r]);
    }
    if (a.out_f32) {
      float* y = reinterpret_cast<float*>(a.y) + o;
      *reinterpret_cast<float4*>(y) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(y + 4) = make_float4(v[4], v[5], v[6], v[7]);
      return u32x4{0u, 0u, 0u, 0u};
    }
    if (a.split_out) {                  // limb layout of the next convolution (split_infer.hip): o = pixel * 3 Co + co, set by the caller
      uint2 h0, l0, h1, l1;
      dh_split4(v, h0, l0);
      dh_split4(v + 4, h1, l1);
      bf16_t* y = reinterpret_cast<bf16_t*>(a.y) + o;
      const u32x4 hi = {h0.x, h0.y, h1.x, h1.y}, lo = {l0.x, l0.y, l1.x, l1.y};
      *reinterpret_cast<u32x4*>(y) = hi;
      *reinterpret_cast<u32x4*>(y + a.Co) = lo;
      *reinterpret_cast<u32x4*>(y + 2 * a.Co) = hi;
      return hi;
    }
    if (a.resid) {
      const bf16_t* rp = reinterpret_cast<const bf16_t*>(&in0);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bf2f(rp[r]);
    }
  } else {
    if (a.mask) {
      const bf16_t* mp = reinterpret_cast<const bf16_t*>(&in0);
#pragma unroll
      for (int r = 0; r < 8; ++r) if (!(bf2f(mp[r]) > 0.f)) v[r] = 0.f;
    }
    if (a.accumulate) {
      const bf16_t* op = reinterpret_cast<const bf16_t*>(&in1);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bf2f(op[r]);
    }
  }
  const u32x4 t = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(a.y) + o) = t;      // (non-temporal stores measured neutral here)
  return t;
}

// ReLU bit mask of 8 packed 16-bit ReLU OUTPUTS (bit r = value r > 0; conv_common.h: dh_pos_bits8_acc)
__device__ __forceinline__ unsigned pos_bits8(const u32x4& t) { return dh_pos_bits8_acc<0>(t, 0u); }
// OR over the four lanes that share frow (the wave's 64 channels of one pixel): lane fq contributes bytes fq (pair 0) and 4 + fq (pair 1)
__device__ __forceinline__ uint2 gather_bits64(unsigned b0, unsigned b1, int fq) {
  return uint2{dh_or_rows(b0 << (8 * fq)), dh_or_rows(b1 << (8 * fq))};
}

// 16-byte LDS-DMA through a buffer descriptor: address = base + voff (per lane) + soff (uniform); a lane whose voff is out
// of range (0xFFFFFFFF) writes ZEROS to its LDS slot.
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// TPS: taps per step (1 or 3); NSW: weight ring depth (prefetch distance D = NSW-1 steps);
// NCU > 0: "thin head" — only the first NCU channel tiles of a wave are computed (Cout <= 16*NCU, ragged Cout allowed).
template <int TH, int TW, int BN, int WM, int WN, int TPS, int NSW, bool DGRAD, int NCU = 0, bool POOL = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo_kernel(const ConvArgs a, const HaloGeom g) {
  constexpr int PW = TW + 2;                       // patch row pitch (pixels); even, so LDS row parity == column parity
  constexpr int PROWS = (TH + 2) * PW;             // patch pixels
  constexpr int PPIECES = (PROWS + 7) / 8;         // 1 KiB DMA pieces per patch (8 pixel rows each)
  constexpr int PBYTES = PPIECES * 1024;
  constexpr int PL = (PPIECES + 7) / 8;            // patch pieces per wave
  constexpr int SPC = 9 / TPS;                     // steps per 64-channel chunk
  constexpr int TAPBYTES = BN * 128;               // weight tile of one tap
  constexpr int WBYTES = TPS * TAPBYTES;           // one ring stage
  constexpr int WL = TPS * BN / 64;                // weight pieces per wave per step
  constexpr int D = NSW - 1;                       // weight prefetch distance (steps)
  constexpr int BM = TH * TW;
  constexpr int TP = BM / WM, TC = BN / WN;        // wave tile: pixels x channels
  constexpr int NPT = TP / 16, NCT = NCU ? NCU : TC / 16;
  static_assert(WM * WN == 8, "8 waves");
  static_assert(TP % 16 == 0 && TC % 16 == 0 && TW % 16 == 0, "MFMA tile alignment");
  static_assert(BN % 64 == 0 && (PW % 2) == 0 && (TPS == 1 || TPS == 3) && NSW >= 3, "layout assumptions");
  static_assert(TPS == 3 ? NSW == 3 : NSW == 4, "stage index arithmetic (SPC = 3 = NSW, or SPC = 9 = 1 mod 4)");
  static_assert(2 * PW * 128 + 64 + PBYTES < 65536, "ds_read immediate offsets");
  static_assert(2 * PBYTES + NSW * WBYTES <= 160 * 1024, "LDS budget");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WRING = 2 * PBYTES;                // LDS map: [patch 0][patch 1][weight ring: NSW stages][bias: Co floats (forward)]
  constexpr int SBIAS = WRING + NSW * WBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int srow = lane >> 3;

  // ---- item mapping --------------------------------------------------------------------------------------------
  const int G = gridDim.x;
  auto decode = [&](int v, int& sp, int& nb) __attribute__((always_inline)) -> bool {     // v = round * G + block
    if (g.grouped) {
      const int r = v / G, b = v - r * G;
      const int xcd = b & 7, slot = b >> 3;
      const int spb = G / g.NB;                            // spatial tiles per round
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

  // ---- patch DMA: per-lane geometry (tile independent) and per-item source offsets ---------------------------------------
  // LDS row R = hy*PW + hx holds pixel (y0-1+hy, x0-1+hx); the 16-byte chunk c of its 64 channels sits at position
  // c ^ (hx & 7).
  int pgeo[PL];                                    // (hy << 8) | hx, or -1 when the slot is beyond the patch
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    int piece = k * 8 + wave;
    if (piece > PPIECES - 1) piece = PPIECES - 1;  // duplicate load of the last piece keeps the per-wave DMA count uniform
    const int row = piece * 8 + srow;
    const int hy = row / PW, hx = row - hy * PW;
    pgeo[k] = row < PROWS ? ((hy << 8) | hx) : -1;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)a.C * 2u), 0x00020000);
  // packed weight rows exist (as zeros) up to the next multiple of 64 output channels (danhip_conv_packed_dims); thin heads read past
  // their 16 rows into the zero fill of the descriptor's range check
  const unsigned wrows = NCU ? (unsigned)a.Co : (unsigned)((a.Co + 63) & ~63);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)(wrows * (unsigned)a.Kpad * 2u), 0x00020000);
  int p_v = blockIdx.x, p_cc = 0, p_idx = 0;       // patch cursor: next chunk to load; p_idx selects the buffer
  int p_sp, p_nb;
  bool p_ok = decode(p_v, p_sp, p_nb);
  unsigned psrc[PL];                               // this lane's byte offset per piece (chunk 0), 0xFFFFFFFF = zero fill
  auto patch_item_setup = [&]() __attribute__((always_inline)) {
    int n, y0, x0;
    sp_coords(p_sp, n, y0, x0);
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int hx = pgeo[k] & 255;
      const int y = y0 - 1 + (pgeo[k] >> 8), x = x0 - 1 + hx;
      const bool ok = pgeo[k] >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const unsigned off = (unsigned)(((n * a.H + y) * a.W + x) * a.C) * 2u + (unsigned)(((lane & 7) ^ (hx & 7)) << 4);
      psrc[k] = ok ? off : 0xFFFFFFFFu;
    }
  };
  auto issue_patch = [&]() __attribute__((always_inline)) {   // loads chunk (p_v, p_cc) into buffer p_idx & 1, advances the cursor
    char* dst = smem + (p_idx & 1) * PBYTES;
    // ragged C (C % 64 != 0, e.g. the data gradient of a 72-channel offsets conv): the last chunk's pieces beyond C are zero-filled,
    // which also cancels whatever the weight tile holds in those K columns (they belong to the next tap)
    const bool ragged_chunk = g.crem != 0 && p_cc == g.cch - 1;
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      int piece = k * 8 + wave;
      if (piece > PPIECES - 1) piece = PPIECES - 1;
      unsigned voff = psrc[k];
      if (ragged_chunk && ((lane & 7) ^ (pgeo[k] & 7)) >= g.crem) voff = 0xFFFFFFFFu;
      bufdma16(rsrc_x, voff, (unsigned)(p_cc * 128), dst + piece * 1024);
    }
    ++p_idx;
    if (++p_cc == g.cch) {
      p_cc = 0;
      p_v += G;
      p_ok = decode(p_v, p_sp, p_nb);
      if (p_ok) patch_item_setup();
    }
  };
  // One-tap-per-step configurations issue ONE piece per step (piece k in step k of the chunk before the one that reads it) instead of all PL
  // in step 0: that step's memory phase was 1230-1350 clocks against ~550 for the others (tools/halo2_trace.hip -DTRACE_HALO1), i.e. one
  // MFMA phase of the partner group lost per chunk.  Piece k is issued AFTER the step's weight pieces, so it is younger than W(c+D) and
  // complete before b1 of step k + D <= 8 -- the barrier after which group B reads the next chunk's first fragments.
  static_assert(SPC != 9 || PL - 1 + D <= SPC - 1, "the last patch piece must have landed before b1 of the chunk's last step");
  auto issue_patch_piece = [&](auto kc) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    char* dst = smem + (p_idx & 1) * PBYTES;
    const bool ragged_chunk = g.crem != 0 && p_cc == g.cch - 1;
    int piece = K * 8 + wave;
    if (piece > PPIECES - 1) piece = PPIECES - 1;
    unsigned voff = psrc[K];
    if (ragged_chunk && ((lane & 7) ^ (pgeo[K] & 7)) >= g.crem) voff = 0xFFFFFFFFu;
    bufdma16(rsrc_x, voff, (unsigned)(p_cc * 128), dst + piece * 1024);
    if (K == PL - 1) {
      ++p_idx;
      if (++p_cc == g.cch) {
        p_cc = 0;
        p_v += G;
        p_ok = decode(p_v, p_sp, p_nb);
        if (p_ok) patch_item_setup();
      }
    }
  };
  // patch pieces of this chunk that may still be in flight at the counted wait of step s: group A has issued steps s-(D-1) .. s after the
  // weight tile it waits for, group B steps s-(D-1) .. s-1
  auto np_a = [](int s) constexpr -> int { int n = 0; for (int k = 0; k < PL; ++k) n += (k >= s - (D - 1) && k <= s) ? 1 : 0; return n; };
  auto np_b = [](int s) constexpr -> int { int n = 0; for (int k = 0; k < PL; ++k) n += (k >= s - (D - 1) && k <= s - 1) ? 1 : 0; return n; };
  if (p_ok) patch_item_setup();

  // ---- weight DMA cursor: step (w_v, w_cc, w_step) to be loaded next; a stage holds TPS taps x BN rows x 128 bytes ---------
  int w_v = blockIdx.x, w_cc = 0, w_step = 0, w_idx = 0;
  int w_sp, w_nb;
  bool w_ok = decode(w_v, w_sp, w_nb);
  // Weight-tile swizzle key of row r (chunk c of row r sits at position c ^ key): the 8 rows one ds_read_b128 group touches
  // must have distinct keys.  Plain kernels interleave pairs of channel tiles (IL: lane row frow of tile c is output channel
  // (c>>1)*32 + (frow>>2)*8 + (c&1)*4 + (frow&3), so a lane owns 8 consecutive channels -> 16-byte epilogue accesses) and
  // read rows {0-3, 8-11 | 16-19, 24-27} (+4 for odd tiles): key = (r & 3) | ((r >> 3) & 1) << 2.  Thin heads read rows 0..15.
  constexpr bool IL = (NCU == 0);
  auto wkey = [](int r) __attribute__((always_inline)) -> int { return IL ? ((r & 3) | (((r >> 3) & 1) << 2)) : (r & 7); };
  unsigned wlane[WL];                              // per piece: ((row in BN) * Kpad + tap_in_step * C) * 2 + swizzled chunk
#pragma unroll
  for (int k = 0; k < WL; ++k) {
    const int pz = wave * WL + k;                  // piece inside the stage
    const int tin = pz / (BN / 8), rp = pz % (BN / 8);
    wlane[k] = (unsigned)((rp * 8 + srow) * a.Kpad + tin * a.C) * 2u + (unsigned)(((lane & 7) ^ wkey(rp * 8 + srow)) << 4);
  }
  auto issue_w = [&]() __attribute__((always_inline)) {
    char* dst = smem + WRING + (w_idx % NSW) * WBYTES;
    const unsigned soff = (unsigned)((w_nb * BN) * a.Kpad + (w_step * TPS) * a.C + w_cc * 64) * 2u;   // uniform
#pragma unroll
    for (int k = 0; k < WL; ++k) bufdma16(rsrc_w, wlane[k], soff, dst + (wave * WL + k) * 1024);
    ++w_idx;
    if (++w_step == SPC) {
      w_step = 0;
      if (++w_cc == g.cch) {
        w_cc = 0;
        w_v += G;
        w_ok = decode(w_v, w_sp, w_nb);
      }
    }
  };

  // ---- ReLU bit mask of the data gradient (ConvArgs::mask_bits, 128-wide tiles): the item's 256 pixels x 16 bytes (128 channels) are
  // DMA'd by waves 0-3 during step 1 of the item's first chunk — after the barriers of step 0, i.e. after every wave's epilogue of the
  // PREVIOUS item has read the buffer — and read from LDS in the epilogue: 4 KiB instead of the 64 KiB bf16 mask tile, off the epilogue's
  // critical path.  The extra DMA only makes the counted waits stricter (one more young operation in the queue of those waves).
  constexpr bool BITS_OK = DGRAD && NCU == 0 && BN == 128 && TPS == 1;
  constexpr bool EMIT_OK = !DGRAD && NCU == 0 && BN == 128;            // forward: writes the bit mask of its own ReLU output (ConvArgs::bits_out)
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.mask_bits ? a.mask_bits : reinterpret_cast<const unsigned char*>(a.x)), 0,
      (int)((unsigned)(a.N * a.H * a.W) * (unsigned)(a.Co / 8)), 0x00020000);
  auto issue_bits = [&](int sp, int nb) __attribute__((always_inline)) {
    int n, y0, x0;
    sp_coords(sp, n, y0, x0);
    const int t = wave * 64 + lane;                                     // (waves 0-3 only)
    const int y = y0 + t / TW, x = x0 + t % TW;
    const unsigned off = (y < a.H && x < a.W) ? (unsigned)((n * a.H + y) * a.W + x) * (unsigned)(a.Co / 8) + (unsigned)(nb * (BN / 8)) : 0xFFFFFFFFu;
    bufdma16(rsrc_b, off, 0u, smem + SBIAS + wave * 1024);
  };

  // ---- compute-side per-lane constants --------------------------------------------------------------------------
  const int frow = lane & 15, fq = lane >> 4;
  const int wrow0 = IL ? ((frow >> 2) * 8 + (frow & 3)) : frow;                   // lane's row inside the first tile (pair)
  const int offW = WRING + (wn * TC + wrow0) * 128 + ((fq ^ wkey(wrow0)) << 4);   // + stage*WBYTES + tap*TAPBYTES + tile offset, ^ ks*64
  int pxaddr[3][NPT];                              // patch-buffer-0 byte address of fragment p at tap column j, k-slice 0
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int t = wm * TP + p * 16 + frow;
    const int ty = t / TW, tx = t % TW;
#pragma unroll
    for (int j = 0; j < 3; ++j) pxaddr[j][p] = (ty * PW + tx + j) * 128 + ((fq ^ ((tx + j) & 7)) << 4);
  }

  f32x4 acc[NCT][NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int p = 0; p < NPT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragments: [slot][k-slice].  TPS == 1: one slot.  TPS == 3: two slots — taps 0 and 1 of the step are read in the mem
  // phase, tap 2 is read at the start of the MFMA phase into slot 0 once tap 0's MFMAs have been issued (its LDS latency
  // hides under tap 1's 16 MFMAs); 96 instead of 144 fragment VGPRs.
  constexpr int NSLOT = TPS == 3 ? 2 : 1;
  bf16x8 wf[NSLOT][2][NCT], xf[NSLOT][2][NPT];
  auto load_tap = [&](auto slotc, auto tapc, int wbase, int pofs) __attribute__((always_inline)) {   // tap TAP (0..8) of the chunk
    constexpr int SLOT = decltype(slotc)::value, TAP = decltype(tapc)::value;
    constexpr int TI = TAP / 3, TJ = TAP % 3, TS = TAP % TPS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int wb = (ks ? (wbase ^ 64) : wbase) + TS * TAPBYTES;
#pragma unroll
      for (int c = 0; c < NCT; ++c) wf[SLOT][ks][c] = *reinterpret_cast<const bf16x8*>(smem + wb + (IL ? (c >> 1) * 4096 + (c & 1) * 512 : c * 2048));
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const int pa = (ks ? (pxaddr[TJ][p] ^ 64) : pxaddr[TJ][p]) + pofs;
        xf[SLOT][ks][p] = *reinterpret_cast<const bf16x8*>(smem + pa + TI * PW * 128);
      }
    }
  };
  auto mma_slot = [&](auto slotc) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slotc)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int p = 0; p < NPT; ++p)
          acc[c][p] = DH_MFMA_16x16x32(wf[SLOT][ks][c], xf[SLOT][ks][p], acc[c][p]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // mem-phase reads of step STEP
  auto load_frags = [&](int wbase, int pofs, auto stepc) __attribute__((always_inline)) {
    constexpr int STEP = decltype(stepc)::value;
    load_tap(I0{}, std::integral_constant<int, STEP * TPS>{}, wbase, pofs);
    if constexpr (TPS == 3) load_tap(I1{}, std::integral_constant<int, STEP * TPS + 1>{}, wbase, pofs);
  };
  // MFMA phase of step STEP (wbase / pofs: for the late tap-2 reads)
  auto mma = [&](int wbase, int pofs, auto stepc) __attribute__((always_inline)) {
    constexpr int STEP = decltype(stepc)::value;
    mma_slot(I0{});
    if constexpr (TPS == 3) {
      __builtin_amdgcn_sched_barrier(0);
      load_tap(I0{}, std::integral_constant<int, STEP * TPS + 2>{}, wbase, pofs);
      __builtin_amdgcn_sched_barrier(0);
      mma_slot(I1{});
      __builtin_amdgcn_sched_barrier(0);
      mma_slot(I0{});
    }
  };

  // ---- prologue -----------------------------------------------------------------------------------------------------
  int c_v = blockIdx.x, c_sp, c_nb;
  bool c_ok = decode(c_v, c_sp, c_nb);
  if (!c_ok) return;                               // (uniform) nothing to do for this block
  issue_patch();
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (w_ok) issue_w();
  // The bias vector lives in LDS for the whole launch: an epilogue that read it from global memory had to wait on vmcnt, i.e. behind
  // every weight / patch DMA already queued for the NEXT steps — one drained prefetch queue per item (measured: forward 13 % slower than
  // the same loop with a store-only epilogue).
  if constexpr (!DGRAD && NCU == 0) {
    float* sb = reinterpret_cast<float*>(smem + SBIAS);
    for (int i = tid; i < a.Co; i += 512) sb[i] = a.bias ? a.bias[i] : 0.f;
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  auto epilogue = [&]() __attribute__((always_inline)) {
    int n, y0, x0;
    sp_coords(c_sp, n, y0, x0);
#ifdef H_ABLATE_EPILOGUE
    // timing-only build (tools/ab_variant.sh with AB_DEFINES=-DH_ABLATE_EPILOGUE): what the item boundary costs - the accumulators are
    // consumed by an empty asm (kept live: cdna_hip_programming.md rule 17) and cleared, nothing is stored
    if constexpr (NCU == 0) {
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          asm volatile("" ::"v"(acc[c][p]));
          acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      c_v += G;
      c_ok = decode(c_v, c_sp, c_nb);
      return;
    }
#endif
    if constexpr (NCU != 0) {                      // thin head: Cout < 64, possibly not a multiple of 4; forward only
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const int t = wm * TP + p * 16 + frow;
        const int y = y0 + t / TW, x = x0 + t % TW;
        const bool ok = y < a.H && x < a.W;
        const size_t m = (size_t)((n * a.H + y) * a.W + x);
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = wn * TC + c * 16 + fq * 4 + r;
            if (ok && co < a.Co) {
              float v = acc[c][p][r] + (a.bias ? a.bias[co] : 0.f);
              if (a.relu) v = dh_relu(v);
              if (a.out_f32) reinterpret_cast<float*>(a.y)[m * a.Co + co] = v;
              else reinterpret_cast<bf16_t*>(a.y)[m * a.Co + co] = f2bf(v);
            }
          }
          acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    } else {
      static_assert(NCU != 0 || NCT % 2 == 0, "channel tiles are stored in interleaved pairs");
      constexpr int NPAIR = NCT / 2 > 0 ? NCT / 2 : 1;
      const int cb = c_nb * BN + wn * TC + fq * 8;    // this lane's 8 consecutive channels of pair 0 (+32 per pair)
      float bv[NPAIR][8];
      bool cok[NPAIR];                               // ragged Cout (Co % 64 != 0, Co % 8 == 0): this lane's 8 channels exist
#pragma unroll
      for (int q = 0; q < NPAIR; ++q) {
        cok[q] = cb + q * 32 < a.Co;
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[q][r] = (!DGRAD && cok[q]) ? reinterpret_cast<const float*>(smem + SBIAS)[cb + q * 32 + r] : 0.f;
      }
      [[maybe_unused]] u32x4 pk[POOL ? NPAIR : 1][POOL ? NPT : 1];      // POOL: packed outputs (zero where the pixel is outside)
      if constexpr (BITS_OK) {
        if (a.mask_bits) {                           // the staged bit mask: 8 bytes = this wave's 64 channels of one pixel
          // The staging area was filled by LDS-DMA in step 1 of this item (retired by the counted waits many steps ago: tests/
          // test_handoff_replay_cpu.py, kind "bits").  Read through the compiler these loads drew an s_waitcnt vmcnt(0) - hipcc orders a
          // ds_read behind every LDS-DMA it has seen - i.e. a full drain of the weight / patch prefetch queue once per item; asm reads with
          // their own lgkmcnt wait do not (cdna_hip_programming.md 5.7, form (ii)).
          dh_u32x2 bbv[NPT];
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            const unsigned la = (unsigned)(size_t)(LDS_AS const char*)(smem + SBIAS + (wm * TP + p * 16 + frow) * 16 + wn * 8);
            asm volatile("ds_read_b64 %0, %1" : "=v"(bbv[p]) : "v"(la) : "memory");
          }
          static_assert(NPT == 4 || NPT == 2, "the wait below names every destination");
          if constexpr (NPT == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bbv[0]), "+v"(bbv[1]), "+v"(bbv[2]), "+v"(bbv[3])::"memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bbv[0]), "+v"(bbv[1])::"memory");
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            const uint2 bb = uint2{bbv[p][0], bbv[p][1]};
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
              const int bi = fq + q * 4;                                // byte of channels cb + q*32 .. +7
              const int byte = (int)((bi < 4 ? bb.x : bb.y) >> (8 * (bi & 3)));
#pragma unroll
              for (int r = 0; r < 8; ++r) {                             // value & (bit ? ~0 : 0): v_bfe_i32 + v_and_b32
                const float v = acc[2 * q + (r >> 2)][p][r & 3];
                acc[2 * q + (r >> 2)][p][r & 3] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe(byte, r, 1));
              }
            }
          }
        }
      }
      if (g.fast) {
        // The common case (everything but fp32 outputs, 16-bit ReLU masks and residual adds): loads / stores through
        // a buffer descriptor with 32-bit offsets (2^31 = out of range = dropped: no exec-mask branch, no 64-bit address arithmetic per
        // store) and no mode branches per fragment.  The epilogue is instruction-issue bound -- both waves of a SIMD run theirs at the
        // same time -- so its length is its instruction count (tools/halo2_trace.hip -DTRACE_HALO1).
        constexpr unsigned OOB = 0x80000000u;
        // (a.y == NULL - the pool-only inference call, danhip_conv2d_fwd_pool(y = NULL) - makes the descriptor ZERO bytes long: every store
        // of the full-resolution map is out of range and dropped by the hardware, without a branch)
#ifdef H_ABLATE_STORES
        // timing-only build: the whole epilogue runs, its full-resolution stores are dropped by a zero-length descriptor
        const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000);
#else
        const __amdgpu_buffer_rsrc_t rsrc_y =
            __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y ? (int)((unsigned)(a.N * a.H * a.W) * (unsigned)a.Co * 2u) : 0, 0x00020000);
#endif
        unsigned pix[NPT];
        bool okp[NPT];
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          const int t = wm * TP + p * 16 + frow;
          const int y = y0 + t / TW, x = x0 + t % TW;
          okp[p] = y < a.H && x < a.W;
          pix[p] = (unsigned)((n * a.H + y) * a.W + x);
        }
        [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrc_bits;
        if constexpr (EMIT_OK) {
          if (a.bits_out)
            rsrc_bits = __builtin_amdgcn_make_buffer_rsrc(a.bits_out, 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)(a.Co / 8)), 0x00020000);
        }
        // ONE optional read-modify input: the gradient an accumulating data gradient adds to.  All 8 loads are requested before the first
        // is used; a lane outside the map or beyond Cout reads zeros (out-of-range offset).  (The 16-bit ReLU mask and the forward's
        // residual take the general epilogue: with a second array live, or this one in the forward instances, the allocator spilled
        // 60-75 registers.)
        const bf16_t* extra = (DGRAD && a.accumulate) ? reinterpret_cast<const bf16_t*>(a.y) : nullptr;      // (uniform)
        [[maybe_unused]] u32x4 inx[NPT][NPAIR];
        if (extra) {
          const __amdgpu_buffer_rsrc_t rsrc_e =
              __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(extra), 0, (int)((unsigned)(a.N * a.H * a.W) * (unsigned)a.Co * 2u), 0x00020000);
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            const unsigned yo = okp[p] ? (pix[p] * (unsigned)a.Co + (unsigned)cb) * 2u : OOB;
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) inx[p][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_e, (int)((cok[q] ? yo : OOB) + q * 64), 0, 0);
          }
        }
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          const unsigned yo = okp[p] ? (pix[p] * (unsigned)a.Co + (unsigned)cb) * 2u : OOB;
          [[maybe_unused]] unsigned pb[2] = {0u, 0u};
          [[maybe_unused]] u32x4 ttq[2];
#pragma unroll
          for (int q = 0; q < NPAIR; ++q) {
            float v[8] = {acc[2 * q][p][0], acc[2 * q][p][1], acc[2 * q][p][2], acc[2 * q][p][3],
                          acc[2 * q + 1][p][0], acc[2 * q + 1][p][1], acc[2 * q + 1][p][2], acc[2 * q + 1][p][3]};
            acc[2 * q][p] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[2 * q + 1][p] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!DGRAD) {
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += bv[q][r];
              if (a.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = dh_relu(v[r]);
              }
            }
            if (extra) {
              const bf16_t* xp = reinterpret_cast<const bf16_t*>(&inx[p][q]);
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += bf2f(xp[r]);
            }
            u32x4 tt = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
#ifdef H_ABLATE_STORE_PATTERN
            // timing-only build (WRONG layout): the same bytes per item, but every store instruction covers ONE contiguous KiB
            // (lane * 16 bytes inside the item's own 64 KiB of the output) instead of 16 half cache lines 512 bytes apart
            __builtin_amdgcn_raw_buffer_store_b128(tt, rsrc_y, (int)((unsigned)((c_sp * g.NB + c_nb) & 0x3fff) * 65536u + (unsigned)(((wave * NPT + p) * NPAIR + q) * 1024 + lane * 16)), 0, 0);
#else
            if constexpr (NPAIR == 2) ttq[q] = tt;
            else __builtin_amdgcn_raw_buffer_store_b128(tt, rsrc_y, (int)((cok[q] ? yo : OOB) + q * 64), 0, 0);
#endif
            if constexpr (POOL) {
              if (!(okp[p] && cok[q])) tt = u32x4{0u, 0u, 0u, 0u};
              pk[q][p] = tt;
            }
            if constexpr (EMIT_OK) {
              if (a.bits_out) pb[q & 1] = pos_bits8(tt);
            }
          }
#ifndef H_ABLATE_STORE_PATTERN
          if constexpr (NPAIR == 2) {
            // WHOLE 128-byte lines per store instruction (round 5; measured in conv_c8.hip and conv_pointwise.hip): the wave's 64 channels are
            // one line of a pixel and a lane holds its two halves (q = 0, 1), so storing piece q from every lane wrote 16 half lines per
            // instruction.  Neighbouring pixels of a row (lane ^ 1) trade pieces: the even pixel's lane ends up with both low halves, the odd
            // one with both high halves; instruction 1 writes the even pixels' lines whole, instruction 2 the odd pixels'.  Same number of
            // store instructions; a pixel outside the map or channels beyond Cout stay out-of-range offsets.
            const bool odd = frow & 1;
            u32x4 give, got;
#pragma unroll
            for (int e = 0; e < 4; ++e) give[e] = odd ? ttq[0][e] : ttq[1][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) got[e] = dh_lane_xor1(give[e]);
            const unsigned yo_n = dh_lane_xor1(yo);              // the neighbour's pixel (OOB when it is outside the map)
            const u32x4 first = odd ? got : ttq[0], second = odd ? ttq[1] : got;
            const unsigned off1 = odd ? (cok[1] ? yo_n : OOB) + 64u : (cok[0] ? yo : OOB);
            const unsigned off2 = odd ? (cok[1] ? yo : OOB) + 64u : (cok[0] ? yo_n : OOB);
            __builtin_amdgcn_raw_buffer_store_b128(first, rsrc_y, (int)off1, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(second, rsrc_y, (int)off2, 0, 0);
          }
#endif
          if constexpr (EMIT_OK) {
            if (a.bits_out) {                        // (uniform) 8 bytes per pixel and wave, stored by the fq == 0 lane
              const uint2 w8 = gather_bits64(pb[0], pb[1], fq);
              const unsigned bo = (okp[p] && fq == 0) ? pix[p] * (unsigned)(a.Co / 8) + (unsigned)(c_nb * (BN / 8) + wn * 8) : OOB;
              __builtin_amdgcn_raw_buffer_store_b64(dh_u32x2{w8.x, w8.y}, rsrc_bits, (int)bo, 0, 0);
            }
          }
        }
        if constexpr (POOL) {
          constexpr int FPR = TW / 16, PV = FPR;
          const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
          const __amdgpu_buffer_rsrc_t rsrc_p =
              __builtin_amdgcn_make_buffer_rsrc(a.pool_y, 0, (int)((unsigned)(a.N * Hp * Wp) * (unsigned)a.Co * 2u), 0x00020000);
          [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrc_pb;
          if constexpr (EMIT_OK) {
            if (a.pool_bits_out)
              rsrc_pb = __builtin_amdgcn_make_buffer_rsrc(a.pool_bits_out, 0, (int)((unsigned)(a.N * Hp * Wp) * (unsigned)(a.Co / 8)), 0x00020000);
          }
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            if (((p / FPR) & 1) != 0) continue;      // top rows of the pairs only
            const int t = wm * TP + p * 16 + frow;
            const int y = y0 + t / TW, x = x0 + t % TW;
            const bool pok = okp[p] && (frow & 1) == 0;
            const unsigned ppix = (unsigned)((n * Hp + (y >> 1)) * Wp + (x >> 1));
            const unsigned po = pok ? (ppix * (unsigned)a.Co + (unsigned)cb) * 2u : OOB;
            [[maybe_unused]] unsigned pb2[2] = {0u, 0u};
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
              u32x4 m;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned v = pkmax_relu(pk[q][p][e], pk[q][p + PV][e]);
                m[e] = pkmax_relu(v, dh_lane_xor1(v));
              }
              __builtin_amdgcn_raw_buffer_store_b128(m, rsrc_p, (int)((cok[q] ? po : OOB) + q * 64), 0, 0);
              if constexpr (EMIT_OK) {
                if (a.pool_bits_out) pb2[q & 1] = pos_bits8(m);
              }
              if (a.pool_arg_out) {                  // (uniform) arg-max codes of the window for danhip_maxpool2x2_bwd_arg: 2 bytes per lane and channel pair
                u32x4 tr;
#pragma unroll
                for (int e = 0; e < 4; ++e) tr[e] = dh_lane_xor1(pk[q][p][e]);
                const unsigned codes = dh_argmax2x2_codes16(pk[q][p], tr, pk[q][p + PV], m);
                if (pok && cok[q])
                  *reinterpret_cast<unsigned short*>(a.pool_arg_out + (size_t)ppix * (a.Co / 4) + (cb + q * 32) / 4) = (unsigned short)codes;
              }
            }
            if constexpr (EMIT_OK) {
              if (a.pool_bits_out) {
                const uint2 w8 = gather_bits64(pb2[0], pb2[1], fq);
                const unsigned bo = (pok && fq == 0) ? ppix * (unsigned)(a.Co / 8) + (unsigned)(c_nb * (BN / 8) + wn * 8) : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(dh_u32x2{w8.x, w8.y}, rsrc_pb, (int)bo, 0, 0);
              }
            }
          }
        }
      } else {
      if constexpr (DGRAD) {
        // Data gradient: the ReLU mask / the value to accumulate into are read-modify inputs from HBM.  ALL of them are issued before the
        // first one is consumed — one exposed memory latency per item instead of one per pixel fragment (the forward epilogue only stores).
        uint4 in0[NPT][NPAIR], in1[NPT][NPAIR];
        bool okp[NPT];
        size_t o0p[NPT];
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          const int t = wm * TP + p * 16 + frow;
          const int y = y0 + t / TW, x = x0 + t % TW;
          okp[p] = y < a.H && x < a.W;
          o0p[p] = (size_t)((n * a.H + y) * a.W + x) * a.Co + cb;
          if (okp[p]) {
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
              if (!cok[q]) continue;
              if (a.mask) in0[p][q] = *reinterpret_cast<const uint4*>(a.mask + o0p[p] + q * 32);
              if (a.accumulate) in1[p][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.y) + o0p[p] + q * 32);
            }
          }
        }
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
#pragma unroll
          for (int q = 0; q < NPAIR; ++q) {
            if (okp[p] && cok[q]) halo_finish8<DGRAD>(a, acc[2 * q][p], acc[2 * q + 1][p], bv[q], in0[p][q], in1[p][q], o0p[p] + q * 32);
            acc[2 * q][p] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[2 * q + 1][p] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      } else {
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const int t = wm * TP + p * 16 + frow;
        const int y = y0 + t / TW, x = x0 + t % TW;
        const bool ok = y < a.H && x < a.W;
        const size_t o0 = (size_t)((n * a.H + y) * a.W + x) * (a.split_out ? 3 * a.Co : a.Co) + cb;
        uint4 in0[NPAIR], in1[NPAIR];              // the residual is read before the first store
        if (ok) {
#pragma unroll
          for (int q = 0; q < NPAIR; ++q) {
            if (!cok[q]) continue;
            if (a.resid && !a.out_f32) in0[q] = *reinterpret_cast<const uint4*>(a.resid + o0 + q * 32);
          }
        }
        [[maybe_unused]] unsigned pb[2] = {0u, 0u};
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
          u32x4 r = {0u, 0u, 0u, 0u};
          if (ok && cok[q]) r = halo_finish8<DGRAD>(a, acc[2 * q][p], acc[2 * q + 1][p], bv[q], in0[q], in1[q], o0 + q * 32);
          if constexpr (POOL) pk[q][p] = r;
          if constexpr (EMIT_OK) pb[q & 1] = pos_bits8(r);
          acc[2 * q][p] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[2 * q + 1][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (EMIT_OK) {
          if (a.bits_out) {                          // (uniform) the ReLU bit mask of y for the consumer's data gradient: 8 bytes per pixel and wave
            const uint2 w8 = gather_bits64(pb[0], pb[1], fq);
            if (ok && fq == 0)
              *reinterpret_cast<uint2*>(a.bits_out + ((size_t)((n * a.H + y) * a.W + x)) * (a.Co / 8) + c_nb * (BN / 8) + wn * 8) = w8;
          }
        }
      }
      }
      if constexpr (POOL) {
        // 2x2 / stride-2 SAME max-pool of this wave's rows (tf.layers.max_pooling2d after the block, net/sfd_net.py:132-143) from
        // the packed ReLU outputs still in registers: the vertical neighbour is fragment p + PV of the same lane, the horizontal
        // one sits in lane ^ 1; even lanes store the pooled pixel (16 bytes per channel pair).
        static_assert((TP / TW) % 2 == 0 && TP / TW >= 2 && TH % 2 == 0 && TW % 2 == 0, "a wave must own whole row pairs");
        constexpr int FPR = TW / 16;                 // fragments per tile row
        constexpr int PV = FPR;                      // fragment index distance of the row below
        const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          if (((p / FPR) & 1) != 0) continue;        // top rows of the pairs only
          const int t = wm * TP + p * 16 + frow;
          const int y = y0 + t / TW, x = x0 + t % TW;
          const bool okp = y < a.H && x < a.W && (frow & 1) == 0;
          const size_t op = ((size_t)((n * Hp + (y >> 1)) * Wp + (x >> 1))) * a.Co + cb;
          [[maybe_unused]] unsigned pb2[2] = {0u, 0u};
#pragma unroll
          for (int q = 0; q < NPAIR; ++q) {
            u32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned v = pkmax_relu(pk[q][p][e], pk[q][p + PV][e]);
              m[e] = pkmax_relu(v, dh_lane_xor1(v));
            }
            if (okp) *reinterpret_cast<u32x4*>(a.pool_y + op + q * 32) = m;
            if constexpr (EMIT_OK) pb2[q & 1] = pos_bits8(m);
            if (a.pool_arg_out) {
              u32x4 tr;
#pragma unroll
              for (int e = 0; e < 4; ++e) tr[e] = dh_lane_xor1(pk[q][p][e]);
              const unsigned codes = dh_argmax2x2_codes16(pk[q][p], tr, pk[q][p + PV], m);
              if (okp && cb + q * 32 < a.Co)
                *reinterpret_cast<unsigned short*>(a.pool_arg_out + ((size_t)((n * Hp + (y >> 1)) * Wp + (x >> 1))) * (a.Co / 4) + (cb + q * 32) / 4) = (unsigned short)codes;
            }
          }
          if constexpr (EMIT_OK) {
            if (a.pool_bits_out) {
              const uint2 w8 = gather_bits64(pb2[0], pb2[1], fq);
              if (okp && fq == 0)
                *reinterpret_cast<uint2*>(a.pool_bits_out + ((size_t)((n * Hp + (y >> 1)) * Wp + (x >> 1))) * (a.Co / 8) + c_nb * (BN / 8) + wn * 8) = w8;
            }
          }
        }
      }
      // Nothing with a register destination may look pending to the compiler when the loop comes round: the residual / mask / old-value
      // loads of THIS path are issued and consumed under conditions hipcc cannot correlate, so it protected their destination registers
      // with an s_waitcnt vmcnt(0) in step 0 of EVERY chunk of every path (in front of group A's fragment reads, in front of group B's
      // barrier) - a full drain of the LDS-DMA queue, and of the lean epilogue's stores, every nine steps.  A builtin wait here (the waitcnt
      // pass understands it; an asm one it would not) ends the uncertainty on the path that caused it.
      __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
      }                                              // (the general epilogue)
    }
    c_v += G;
    c_ok = decode(c_v, c_sp, c_nb);
  };

  // ---- main loop ----------------------------------------------------------------------------------------------------
  // Cycle c = step c of this block's flattened (item, chunk, step) sequence.  Hand-offs, all checked at b1 of cycle c
  // (after which B reads the tile of step c+1):
  //   * every wave has waited for its own pieces of weight tile c+1 (both groups issue tile c+D during cycle c);
  //   * stage (c+D) % NSW == (c-1) % NSW was last read by A in mem(c-1) and by B in the mem phase of cycle c-2;
  //   * the patch of chunk q+1 is issued at step 0 of chunk q and retired by the counted waits long before its first use.
  // ONE barrier per cycle carries every hand-off (b1).  The second one (b2, after A's MFMA phase / B's memory phase; option "halo_b2") only
  // forced the two groups' phases to alternate strictly, and cost 5-7 % (same-box A/B, profiles/r3/README.md): the alternation survives
  // without it because each group's own program order is mem -> MFMA -> mem ..., re-aligned at every b1.  Why b1 alone is enough:
  //   * slot (c+D) % NSW is written from cycle c on (A: in mem(c), i.e. after passing b1(c-1); B: after b1(c)).  Its last readers read
  //     tile c-1: A in mem(c-1), complete (lgkmcnt(0)) before A arrives at b1(c-1); B in its memory phase of cycle c-2, complete before
  //     B's MFMA phase of cycle c-1, i.e. before B arrives at b1(c-1).  A wave that has PASSED b1(c-1) knows every wave has arrived there.
  //   * the same argument with chunks for the two patch buffers (written from step 0 of chunk q, last read for the last step of chunk
  //     q-1) and for the ReLU-bit staging area (written in step 1 of an item, last read by the previous item's epilogues: A's in
  //     mem(0), B's before its MFMA phase of step 0 -- both before b1 of step 0).
  // This holds because NO LDS read happens after b1 of its cycle.  The three-taps-per-step form (TPS == 3) reads its third tap's fragments
  // INSIDE the MFMA phase: without b2 a faster wave of group A could start mem(c+1) -- and rewrite the slot / patch buffer of step c-1..c
  // -- while a slower A wave is still reading tap 2 of step c.  Those instances keep the second barrier.
  // DMA queue of a wave, oldest first: A at its wait in cycle c: [W(c+1)] ... [W(c+D)] -> (D-1)*WL may stay in flight, plus
  // a patch if it was issued within the last D-1 cycles; B (which has issued up to cycle c-1): (D-2)*WL, patch age <= D-2.
  int chunk = 0, cc = 0;                           // running chunk number: patch buffer = chunk & 1
  [[maybe_unused]] int tr_step = 0;
  int patch_age = 16;                              // cycles since this wave last issued a patch (large: outside every window)
  bool p_live = false;                             // one-tap-per-step form: this chunk issues patch pieces (set in its step 0)
  auto stage_of = [&](int step) __attribute__((always_inline)) -> int {
    return TPS == 3 ? step : (chunk + step) & (NSW - 1);      // TPS == 3: SPC == NSW; TPS == 1: SPC = 9 = 1 (mod 4)
  };
  using S0 = std::integral_constant<int, 0>;

  if (wave < 4) {
    // ================================================= group A =================================================
    // An item that finished in the previous cycle has its epilogue at the top of the loop: ONE inlined copy for both the "next item follows"
    // and the "last item of this block" case (the epilogue is ~700-1600 instructions and the kernel must stay inside the 64 KB I-cache).
    bool pending = false, done = false;
    for (;;) {
      if (pending) {
        epilogue();
        pending = false;
        if (done) break;
        __builtin_amdgcn_sched_barrier(0);
      }
      const int pofs = (chunk & 1) * PBYTES;
      auto cycle = [&](auto stepc) __attribute__((always_inline)) {
        constexpr int STEP = decltype(stepc)::value;
        // ---- mem phase: fragment reads first (their LDS latency runs under the epilogue / DMA issue below)
        const int wbase = offW + stage_of(STEP) * WBYTES;
        H_STAMP(0);
        load_frags(wbase, pofs, stepc);
        __builtin_amdgcn_sched_barrier(0);
        const bool more_w = w_ok;
        if (more_w) issue_w();
        if constexpr (SPC == 9) {
          if (STEP == 0) p_live = p_ok;
          if constexpr (STEP < PL) {
            if (p_live) issue_patch_piece(stepc);
          }
        } else {
          if (STEP == 0 && p_ok) { issue_patch(); patch_age = 0; }
        }
        if constexpr (BITS_OK) {
          if (STEP == 1 && cc == 0 && a.mask_bits) issue_bits(c_sp, c_nb);
        }
        if (!more_w) wait_vmcnt<0>();              // tail of this block's work
        else if constexpr (SPC == 9) {
          if (p_live) wait_vmcnt<(D - 1) * WL + np_a(STEP)>();
          else wait_vmcnt<(D - 1) * WL>();
        } else {
          if (patch_age <= D - 1) wait_vmcnt<(D - 1) * WL + PL>();
          else wait_vmcnt<(D - 1) * WL>();
        }
        ++patch_age;
        __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): fragments are in registers before the MFMA phase starts
        H_STAMP(1);
        __builtin_amdgcn_s_barrier();              // b1
        H_STAMP(2);
        // ---- MFMA phase
        __builtin_amdgcn_sched_barrier(0);
        mma(wbase, pofs, stepc);
        __builtin_amdgcn_sched_barrier(0);
        H_STAMP(3);
        if (TPS == 3 || g.b2) __builtin_amdgcn_s_barrier();      // b2 (always with three taps per step: see the hand-off notes)
        H_STEP_DONE();
      };
      cycle(std::integral_constant<int, 0>{});
      cycle(std::integral_constant<int, 1>{});
      cycle(std::integral_constant<int, 2>{});
      if constexpr (SPC == 9) {
        cycle(std::integral_constant<int, 3>{});
        cycle(std::integral_constant<int, 4>{});
        cycle(std::integral_constant<int, 5>{});
        cycle(std::integral_constant<int, 6>{});
        cycle(std::integral_constant<int, 7>{});
        cycle(std::integral_constant<int, 8>{});
      }
      ++chunk;
      if (++cc == g.cch) {
        cc = 0;
        int nsp, nnb;
        pending = true;
        done = !decode(c_v + G, nsp, nnb);         // the block's last item
      }
    }
  } else {
    // ================================================= group B =================================================
    bool w_prev = true;                            // did the previous cycle issue a weight tile
    load_frags(offW, 0, S0{});
    for (;;) {
      const int pofs = (chunk & 1) * PBYTES;
      bool last = false;                           // set when the block's last item has been finished
      auto cycle = [&](auto stepc) __attribute__((always_inline)) {
        constexpr int STEP = decltype(stepc)::value;
        // ---- MFMA phase (step c)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        H_STAMP(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(offW + stage_of(STEP) * WBYTES, pofs, stepc);
        __builtin_amdgcn_sched_barrier(0);
        H_STAMP(1);
        if (!w_prev) wait_vmcnt<0>();
        else if constexpr (SPC == 9) {
          if (p_live) wait_vmcnt<(D - 2) * WL + np_b(STEP)>();
          else wait_vmcnt<(D - 2) * WL>();
        } else {
          if (patch_age <= D - 2) wait_vmcnt<(D - 2) * WL + PL>();
          else wait_vmcnt<(D - 2) * WL>();
        }
        __builtin_amdgcn_s_barrier();              // b1
        H_STAMP(2);
        // ---- mem phase (for step c+1): reads first
        constexpr int NSTEP = (STEP + 1) % SPC;
        const int nstage = TPS == 3 ? NSTEP : (chunk + STEP + 1) & (NSW - 1);
        const int npofs = STEP == SPC - 1 ? (PBYTES - pofs) : pofs;
        if constexpr (DGRAD) {
          if (STEP == SPC - 1 && cc + 1 == g.cch) {
            epilogue();
            if (!c_ok) last = true;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        load_frags(offW + nstage * WBYTES, npofs, std::integral_constant<int, NSTEP>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!DGRAD) {
          if (STEP == SPC - 1 && cc + 1 == g.cch) {  // the item ended with this step
            epilogue();
            if (!c_ok) last = true;
          }
        }
        const bool more_w = w_ok;
        if (more_w) issue_w();
        w_prev = more_w;
        ++patch_age;
        if constexpr (SPC == 9) {
          if (STEP == 0) p_live = p_ok;
          if constexpr (STEP < PL) {
            if (p_live) issue_patch_piece(stepc);
          }
        } else {
          if (STEP == 0 && p_ok) { issue_patch(); patch_age = 0; }
        }
        H_STAMP(3);
        if (TPS == 3 || g.b2) __builtin_amdgcn_s_barrier();      // b2 (always with three taps per step: see the hand-off notes)
        H_STEP_DONE();
      };
      cycle(std::integral_constant<int, 0>{});
      cycle(std::integral_constant<int, 1>{});
      cycle(std::integral_constant<int, 2>{});
      if constexpr (SPC == 9) {
        cycle(std::integral_constant<int, 3>{});
        cycle(std::integral_constant<int, 4>{});
        cycle(std::integral_constant<int, 5>{});
        cycle(std::integral_constant<int, 6>{});
        cycle(std::integral_constant<int, 7>{});
        cycle(std::integral_constant<int, 8>{});
      }
      ++chunk;
      if (++cc == g.cch) cc = 0;
      if (last) break;
    }
  }
#ifdef H_TRACE
  wait_vmcnt<0>();
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = tid; i < 1024; i += 512) g.trace[i] = reinterpret_cast<const unsigned*>(smem + SBIAS + 4096)[i];
#endif
}

#ifdef H_TRACE
unsigned* h_trace_buffer() {
  static unsigned* p = [] { void* q = nullptr; (void)hipMalloc(&q, 4096 + 64); (void)hipMemset(q, 0, 4096 + 64); return (unsigned*)q; }();
  return p;
}
#endif

struct HaloPlan {
  int th, tw, bn;
  int head;      // thin head: Cout <= 16, one channel tile
};

// Picks the tile; returns false when the shape should go to the flat-M kernel.
bool plan_halo(const ConvArgs& a, HaloPlan* p) {
  if (!(a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dstride == 1 && a.pad_t == 1 && a.pad_l == 1)) return false;
  if (a.H != a.Ho || a.W != a.Wo) return false;
  p->head = 0;
  if (a.C % 8 != 0 || a.C < 64 || a.C > 2048) return false;               // C % 64 != 0: the last chunk is zero-filled beyond C
  if (a.Co % 64 != 0) {
    if (a.Co <= 16) {
      if (a.mask || a.resid || a.accumulate) return false;                 // thin head (forward only)
      p->head = 1;
    } else if (a.Co % 8 != 0 || a.Co <= 32) return false;                  // ragged Cout: whole 8-channel groups, packed rows padded to 64
  }
  if ((int64_t)a.Co * a.Kpad >= (1ll << 31) || a.Co > 2048) return false;          // (the forward keeps Co bias floats in 8 KiB of LDS)
  auto util = [&](int th, int tw) {
    const double ph = (double)((a.H + th - 1) / th * th), pw = (double)((a.W + tw - 1) / tw * tw);
    return (double)a.H * a.W / (ph * pw);
  };
  const double u1 = util(8, 32), u2 = util(16, 16);
  if (u1 >= u2) { p->th = 8; p->tw = 32; } else { p->th = 16; p->tw = 16; }
  if ((u1 > u2 ? u1 : u2) < 0.78) return false;      // 40x40 maps (0.69): measured equal to the flat-M kernel (tile + round quantisation)
  p->bn = p->head ? 64 : ((((a.Co + 63) & ~63) % 128 == 0) ? 128 : 64);
  return true;
}

int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

template <int TH, int TW, int BN, int WM, int WN, int TPS, int NSW, bool DGRAD, int NCU = 0, bool POOL = false>
int launch_halo_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int PPIECES = ((TH + 2) * (TW + 2) + 7) / 8;
  constexpr int LDS0 = 2 * PPIECES * 1024 + NSW * TPS * BN * 128;
  // + the bias vector (forward, Co <= 2048 floats; <= 512 in the 3-taps-per-phase 64-wide form, which fills the LDS) / the item's 4 KiB ReLU bit mask
  constexpr int LDS = LDS0 + (NCU == 0 ? (TPS == 3 ? 2048 : 8192) : 0);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<TH, TW, BN, WM, WN, TPS, NSW, DGRAD, NCU, POOL>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  HaloGeom g{};
  g.tiles_x = (a.W + TW - 1) / TW;
  g.tiles_y = (a.H + TH - 1) / TH;
  g.sp_items = a.N * g.tiles_x * g.tiles_y;
  g.NB = NCU ? 1 : (a.Co + BN - 1) / BN;
  g.cch = (a.C + 63) / 64;
  g.crem = (a.C % 64) / 8;
  g.b2 = danhip_option("halo_b2");
  g.fast = (NCU == 0 && !a.out_f32 && !a.split_out && !a.mask && !a.resid && (int64_t)a.N * a.H * a.W * a.Co * 2 <= (1ll << 31) &&
            (!danhip_option("halo_general_epilogue") || !a.y)) ? 1 : 0;
  if (!a.y && !(g.fast && POOL)) { danhip_set_error("conv_halo: y == NULL (pool-only) needs the pool-fusing instance with the lean epilogue"); return DANHIP_EINVAL; }
#ifdef H_TRACE
  g.trace = h_trace_buffer();
#endif
  g.div_tx = make_fastdiv(g.tiles_x);
  g.div_txy = make_fastdiv(g.tiles_x * g.tiles_y);
  g.div_nb = make_fastdiv(g.NB);
  const long items = (long)g.sp_items * g.NB;
  int G = cu_count();
  g.grouped = 0;
  if (items >= G && (G % 8) == 0 && ((G / 8) % g.NB) == 0 && g.NB > 1) g.grouped = 1;
  if (items < G) G = (int)items;
  hipLaunchKernelGGL((conv3x3_halo_kernel<TH, TW, BN, WM, WN, TPS, NSW, DGRAD, NCU, POOL>), dim3(G), dim3(512), LDS, s, a, g);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

// configurations: <TH, TW, BN, WM, WN, TPS, NSW>
//   128-wide : 4x2 waves (64 px x 64 co each), 1 tap / step, 4-deep ring
//    64-wide : 8x1 waves (32 px x 64 co each), 1 tap / step, 4-deep ring.  (The 3-taps-per-step form <.., 3, 3> is correct
//              and kept instantiable, but measured slower on conv1_2 / conv2_1-dgrad: with one or two chunks per item the
//              per-item epilogue + patch DMA land in one long mem phase.)
template <bool DGRAD>
int launch_halo(const ConvArgs& a, const HaloPlan& p, hipStream_t s) {
  if (p.head) {
    if (DGRAD) return 1;
    return p.th == 8 ? launch_halo_cfg<8, 32, 64, 8, 1, 3, 3, false, 1>(a, s) : launch_halo_cfg<16, 16, 64, 8, 1, 3, 3, false, 1>(a, s);
  }
  if constexpr (!DGRAD) {
    if (a.pool_y && p.bn == 128)                   // fused 2x2 max-pool epilogue (the 128-wide tiles own whole row pairs per wave)
      return p.th == 8 ? launch_halo_cfg<8, 32, 128, 4, 2, 1, 4, false, 0, true>(a, s) : launch_halo_cfg<16, 16, 128, 4, 2, 1, 4, false, 0, true>(a, s);
  }
  if (p.th == 8) {
    static const int exp3 = [] { const char* e = getenv("DANHIP_HALO_TPS3"); return e ? atoi(e) : 0; }();      // experiment: 64-wide tiles, 3 taps per phase
    if (exp3 && !a.mask_bits && !a.bits_out && !a.pool_y && a.Co <= 512) return launch_halo_cfg<8, 32, 64, 8, 1, 3, 3, DGRAD>(a, s);
    if (p.bn == 128) return launch_halo_cfg<8, 32, 128, 4, 2, 1, 4, DGRAD>(a, s);
    return launch_halo_cfg<8, 32, 64, 8, 1, 1, 4, DGRAD>(a, s);
  }
  if (p.bn == 128) return launch_halo_cfg<16, 16, 128, 4, 2, 1, 4, DGRAD>(a, s);
  return launch_halo_cfg<16, 16, 64, 8, 1, 1, 4, DGRAD>(a, s);
}

}  // namespace

// dgrad mode = no bias / relu / residual / fp32 output requested (the data-gradient call); forward otherwise.
int danhip_launch_conv_halo(const ConvArgs& a, hipStream_t s) {
  HaloPlan p;
  if (!plan_halo(a, &p)) return 1;
  const bool dgrad = !a.bias && !a.relu && !a.resid && !a.out_f32 && !a.split_out && !p.head;
  if (a.split_out && p.head) return 1;
  if (a.mask_bits && !(dgrad && p.bn == 128 && a.Co % 128 == 0)) return 1;
  if (!dgrad && a.accumulate) return 1;
  if (!dgrad && a.mask) return 1;
  return dgrad ? launch_halo<true>(a, p, s) : launch_halo<false>(a, p, s);
}

bool danhip_conv_halo_emits_bits(const ConvArgs& a) {
  HaloPlan p;
  if (!plan_halo(a, &p) || p.head || p.bn != 128 || a.Co % 128 != 0) return false;
  return a.bias && a.relu && !a.resid && !a.out_f32 && !a.mask && !a.accumulate;      // forward conv_relu
}

bool danhip_conv_halo_takes_bits(const ConvArgs& a) {
  HaloPlan p;
  if (!plan_halo(a, &p) || p.head || p.bn != 128 || a.Co % 128 != 0) return false;
  return !a.bias && !a.relu && !a.resid && !a.out_f32;                 // the data-gradient form
}

bool danhip_conv_halo_pool_fusable(const ConvArgs& a) {
  HaloPlan p;
  if (!plan_halo(a, &p) || p.head || p.bn != 128 || a.Co % 64 != 0) return false;
  return a.bias && a.relu && !a.resid && !a.out_f32 && !a.mask && !a.accumulate;      // forward conv_relu only
}

const char* danhip_conv_halo_label(const ConvArgs& a, bool dgrad) {
  HaloPlan p;
  if (!plan_halo(a, &p)) return nullptr;
  if (p.head) {
    if (dgrad) return nullptr;
    return p.th == 8 ? "conv3x3_halo_kernel<8, 32, 64, 8, 1, 3, 3, false, 1, false>" : "conv3x3_halo_kernel<16, 16, 64, 8, 1, 3, 3, false, 1, false>";
  }
  if (!dgrad && a.pool_y && danhip_conv_halo_pool_fusable(a))
    return p.th == 8 ? "conv3x3_halo_kernel<8, 32, 128, 4, 2, 1, 4, false, 0, true>" : "conv3x3_halo_kernel<16, 16, 128, 4, 2, 1, 4, false, 0, true>";
  if (p.th == 8) {
    if (p.bn == 128) return dgrad ? "conv3x3_halo_kernel<8, 32, 128, 4, 2, 1, 4, true, 0, false>" : "conv3x3_halo_kernel<8, 32, 128, 4, 2, 1, 4, false, 0, false>";
    return dgrad ? "conv3x3_halo_kernel<8, 32, 64, 8, 1, 1, 4, true, 0, false>" : "conv3x3_halo_kernel<8, 32, 64, 8, 1, 1, 4, false, 0, false>";
  }
  if (p.bn == 128) return dgrad ? "conv3x3_halo_kernel<16, 16, 128, 4, 2, 1, 4, true, 0, false>" : "conv3x3_halo_kernel<16, 16, 128, 4, 2, 1, 4, false, 0, false>";
  return dgrad ? "conv3x3_halo_kernel<16, 16, 64, 8, 1, 1, 4, true, 0, false>" : "conv3x3_halo_kernel<16, 16, 64, 8, 1, 1, 4, false, 0, false>";
}
