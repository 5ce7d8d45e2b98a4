// Pointwise (1x1, stride 1) convolution = streaming GEMM on MFMA for gfx950:  Y[M, Co] = X[M, C] . Wp[Co, C]^T  (+ bias, ReLU), and
// run on dY with the transposed packing its data gradient  dX[M, Cin] (+)= dY[M, Co8] . Wb[Cin, Co8]^T  (* (x > 0), + old).
//
// Reference call sites: the LFPN lateral / upsample convs (net/pb_net.py:198-218, net/danet.py:352-372), every 1x1 of the DAN context
// modules (net/danet.py:842-918, net/danet_deform.py:267-290), the stage-2 mixing convs (net/danet.py:944-947), fc7 / conv6_1 / conv7_1
// (net/sfd_net.py:146-154), and the GEMM half of DeformConvOp over its sampled columns (cpp/Deform/deform_conv.cc:509-518: K = 9 C).
// Round 1 sent these shapes to hipBLASLt; this kernel replaces the library on the default path.  With TAPS it is also the implicit-GEMM
// kernel of every window convolution the halo kernels do not take (maps <= 40 px wide, stride 2, 3x1 / 1x3), in place of the two-stage
// flat-M kernel (conv_igemm.hip keeps the ragged-channel shapes and the tiny maps).
//
// These products are HBM-bound (read X once, write Y once; the weights are L2-resident): the design is a streaming one.
//  * persistent 512-thread workgroup per CU, tile = 128 pixels x BN output channels (BN = 256 / 128 / 64 = the whole Co for the
//    common layers, so X is read exactly once), K walked in 64-channel steps through an NST-deep LDS ring;
//  * both operands are K-contiguous in HBM ([pixel][C] NHWC rows, packed [Co][Kpad] weights): 16-byte LDS-DMA pieces of 8 rows x 128 B,
//    XOR-swizzled on the source side, ds_read_b128 fragments, v_mfma_f32_16x16x32 with the weight fragment as the A operand (a lane
//    owns 8 consecutive output channels of one pixel -> 16-byte NHWC stores, channel tiles interleaved in pairs as in conv_halo.hip);
//  * the step sequence is flattened across items: the next item's first K-steps are already landing while the epilogue stores run;
//    the DMA is inline asm with a counted vmcnt (never drained in the loop: hipcc would wait vmcnt(0) before each ds_read), the
//    epilogue's mask / old-value reads are asm loads issued BEFORE the item's last DMA so that waiting for them retires nothing else;
//  * XCD-grouped item order: each XCD walks a contiguous eighth of the (pixel tile, column block) items, so the Co / BN column blocks of a
//    pixel tile run on one XCD (second .. last read X from that L2) and, with taps, neighbouring pixel tiles' halo rows meet there too.
#include <type_traits>

#include "conv_common.h"

namespace {

struct PwGeom {
  int m_tiles, NB, items, ksteps, grouped, per_xcd;
  FastDiv div_nb;
};

typedef __attribute__((ext_vector_type(4))) unsigned pw_u32x4;

template <int N>
__device__ __forceinline__ void pw_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pw_dma16(pw_u32x4 rsrc, unsigned voff, unsigned lds_addr) {      // voff out of range: zeros land in LDS
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ pw_u32x4 pw_make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return pw_u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
__device__ __forceinline__ pw_u32x4 pw_load16(const void* p) {        // asm load: counted by hand (pw_wait_vmcnt + pw_land)
  pw_u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void pw_land(pw_u32x4& v) { asm volatile("" : "+v"(v)); }

// Dynamic wait: at most `n` vector-memory operations of this wave may stay outstanding (n wave-uniform, clamped to the hardware's 63)
__device__ __forceinline__ void pw_wait_vmcnt_dyn(int n) {
  if (n <= 0) pw_wait_vmcnt<0>();
  else if (n <= 2) pw_wait_vmcnt<2>();
  else if (n <= 4) pw_wait_vmcnt<4>();
  else if (n <= 6) pw_wait_vmcnt<6>();
  else if (n <= 8) pw_wait_vmcnt<8>();
  else if (n <= 12) pw_wait_vmcnt<12>();
  else if (n <= 16) pw_wait_vmcnt<16>();
  else if (n <= 24) pw_wait_vmcnt<24>();
  else pw_wait_vmcnt<32>();
}
// ... rounded DOWN to the nearest encodable step (waiting for more than necessary is always safe)
__device__ __forceinline__ int pw_floor_count(int n) {
  return n >= 32 ? 32 : n >= 24 ? 24 : n >= 16 ? 16 : n >= 12 ? 12 : n >= 8 ? 8 : n >= 6 ? 6 : n >= 4 ? 4 : n >= 2 ? 2 : 0;
}

// LD: the epilogue has inputs to read (forward: bias; data gradient: ReLU mask and / or the old value to accumulate onto)
// TAPS: a kh x kw window (any stride for the forward pass, stride 1 for the data gradient) instead of 1x1 — the implicit-GEMM form:
// K-step (tap, 64-channel chunk) gathers the tap's shifted pixel rows; padding pixels are out-of-range lanes (zero fill).  This is what
// carries the 3x3 convolutions on maps too small for the halo tiles (40x40, 20x20: conv5_x, fc6, the CPM / context levels 2..5), the
// stride-2 extra layers and DAN's 3x1 / 1x3 branches.
// TWO (round 4; forward, 1x1 only): the K axis is the concatenation of TWO source tensors of equal pitch - K-steps [0, a.ksplit) read a.x,
// the rest a.x2 - i.e. Y = relu([X1 | X2] . W^T + b) without materialising the concatenation (DAN's stage-2 input mix: net/danet.py:944-950)
template <int BN, int NST, bool DGRAD, bool LD, bool TAPS, bool TWO>
__device__ __forceinline__ void pw_body(const ConvArgs a, const PwGeom g) {
  static_assert(!TWO || (!TAPS && !DGRAD), "two-source K: plain forward 1x1 only");
  constexpr int BM = 128;
  constexpr int WN = BN / 64, WM = 8 / WN;            // waves along Co / along pixels
  constexpr int TP = BM / WM;                         // pixels per wave (64 / 32 / 16)
  constexpr int NPT = TP / 16, NCT = 4, NPAIR = 2;    // wave tile: TP pixels x 64 channels
  constexpr int ABYTES = BM * 128, WBYTES = BN * 128, SB = ABYTES + WBYTES;
  constexpr int APW = 2, WPW = BN / 64;               // DMA pieces per wave and K-step (A: 16 pieces, W: BN / 8 pieces)
  constexpr int DPW = APW + WPW;
  static_assert(NST * SB <= 160 * 1024, "LDS budget");
  static_assert((NST - 1) * DPW + 2 * NPT * NPAIR <= 63, "vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int srow = lane >> 3;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS char*)smem);
  const int G = gridDim.x;

  auto decode = [&](int v, int& mt, int& nb) __attribute__((always_inline)) -> bool {     // v = round * G + block
    int it;
    if (g.grouped) {
      // an XCD (blocks b, b + 8, ...) walks its own contiguous EIGHTH of the items, G / 8 at a time (round 5; rounds 2-4: a contiguous run
      // per round, and only when G / 8 was a multiple of NB - Co / BN = 9, the deformable column gradient, fell back to item = block and
      // its nine column blocks of a pixel tile ran on nine XCDs: X fetched ~9 x, rocprofv3 FETCH_SIZE)
      const int r = v / G, b = v - r * G;
      const int j = r * (G >> 3) + (b >> 3);
      it = (b & 7) * g.per_xcd + j;
      if (j >= g.per_xcd) it = g.items;                // this XCD's share is done
    } else {
      it = v;
    }
    mt = (int)fdiv((unsigned)it, g.div_nb);
    nb = it - mt * g.NB;
    return it < g.items;
  };

  // ---- DMA geometry.  LDS rows are 128 bytes (64 channels); chunk c of row r sits at position c ^ key(r).
  // A rows (pixels): key = r & 7.  W rows: pairs of channel tiles are interleaved (lane row frow of tile c is output channel
  // (c>>1)*32 + (frow>>2)*8 + (c&1)*4 + (frow&3)), a ds_read_b128 group touches rows {0-3, 8-11 | 16-19, 24-27}: key = (r&3) | ((r>>3)&1)<<2.
  auto wkey = [](int r) __attribute__((always_inline)) -> int { return (r & 3) | (((r >> 3) & 1) << 2); };
  // (pitched view: the last pixel's row ends C elements after its start, whatever the pitch)
  const unsigned c_first = TWO ? (unsigned)a.ksplit * 64u : (unsigned)a.C;         // channels the first source carries
  const pw_u32x4 rsrc_x = pw_make_rsrc(a.x, ((unsigned)(a.N * a.H * a.W - 1) * (unsigned)a.ldx + c_first) * 2u);      // rows >= M are out of range: zero fill
  [[maybe_unused]] const pw_u32x4 rsrc_x2 =
      pw_make_rsrc(TWO ? a.x2 : a.x, ((unsigned)(a.N * a.H * a.W - 1) * (unsigned)a.ldx + ((unsigned)a.C - c_first)) * 2u);
  const pw_u32x4 rsrc_w = pw_make_rsrc(a.w, (unsigned)(g.NB * BN) * (unsigned)a.Kpad * 2u);
  unsigned avoff[APW], wvoff[WPW];
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    const int row = (wave * APW + k) * 8 + srow;
    avoff[k] = (unsigned)row * (unsigned)(a.ldx * 2) + (unsigned)(((lane & 7) ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int k = 0; k < WPW; ++k) {
    const int row = (wave * WPW + k) * 8 + srow;
    wvoff[k] = (unsigned)row * (unsigned)(a.Kpad * 2) + (unsigned)(((lane & 7) ^ wkey(row)) << 4);
  }
  int d_v = blockIdx.x, d_k = 0, d_idx = 0, d_mt, d_nb;
  bool d_ok = decode(d_v, d_mt, d_nb);
  // TAPS: per staged pixel row the byte offset of its tap-(0,0) source pixel (+ swizzled chunk) and a bit mask of the taps that fall
  // inside the image, rebuilt when the DMA cursor enters a new item; a K-step adds the wave-uniform (tap, chunk) offset
  [[maybe_unused]] unsigned pbase[APW], tmask[APW];
  [[maybe_unused]] int d_ti = 0, d_tj = 0, d_cc = 0, d_tap = 0;
  auto item_rows = [&]() __attribute__((always_inline)) {
    if constexpr (TAPS) {
#pragma unroll
      for (int k = 0; k < APW; ++k) {
        const int row = (wave * APW + k) * 8 + srow;
        const int m = d_mt * BM + row;
        unsigned mk = 0;
        unsigned pb = 0;
        if (d_ok && m < a.M) {
          const unsigned n = fdiv((unsigned)m, a.div_howo);
          const unsigned rem = (unsigned)m - n * (unsigned)(a.Ho * a.Wo);
          const unsigned ho = fdiv(rem, a.div_wo);
          const int rh = (int)ho * a.stride - a.pad_t, rw = (int)(rem - ho * (unsigned)a.Wo) * a.stride - a.pad_l;
          pb = (unsigned)((((int)n * a.H + rh) * a.W + rw) * a.ldx) * 2u + (unsigned)(((lane & 7) ^ (row & 7)) << 4);   // may wrap; exact for valid taps
          for (int t = 0; t < a.taps; ++t) {
            const int i = (int)fdiv((unsigned)t, a.div_kw), jj = t - i * a.kw;
            if ((unsigned)(rh + i) < (unsigned)a.H && (unsigned)(rw + jj) < (unsigned)a.W) mk |= 1u << t;
          }
        }
        pbase[k] = pb;
        tmask[k] = mk;
      }
    }
  };
  item_rows();
  auto issue = [&]() __attribute__((always_inline)) {      // K-step (d_v, d_k) into ring slot d_idx % NST; always DPW instructions
    const unsigned base = lds0 + (unsigned)(d_idx % NST) * SB;
    // (the whole offset goes through the per-lane operand: that one is range-checked, so pixel rows >= M and the steps beyond the
    // stream read zeros)
    const unsigned inv = d_ok ? 0u : 0xFFFFFFFFu;
    const unsigned sw = (unsigned)(d_nb * BN) * (unsigned)(a.Kpad * 2) + (unsigned)(d_k * 128);
    if constexpr (TAPS) {
      const unsigned toff = (unsigned)((d_ti * a.W + d_tj) * a.ldx + d_cc * 64) * 2u;       // wave-uniform
#pragma unroll
      for (int k = 0; k < APW; ++k) {
        const unsigned voff = ((tmask[k] >> d_tap) & 1u) ? pbase[k] + toff : 0xFFFFFFFFu;
        pw_dma16(rsrc_x, voff | inv, base + (wave * APW + k) * 1024);
      }
    } else {
      unsigned kk = (unsigned)d_k;
      pw_u32x4 rs = rsrc_x;
      if constexpr (TWO) {
        if (d_k >= a.ksplit) { kk = (unsigned)(d_k - a.ksplit); rs = rsrc_x2; }      // wave-uniform: four s_cselect
      }
      const unsigned sa = (unsigned)(d_mt * BM) * (unsigned)(a.ldx * 2) + kk * 128u;
#pragma unroll
      for (int k = 0; k < APW; ++k) pw_dma16(rs, (avoff[k] + sa) | inv, base + (wave * APW + k) * 1024);
    }
#pragma unroll
    for (int k = 0; k < WPW; ++k) pw_dma16(rsrc_w, (wvoff[k] + sw) | inv, base + ABYTES + (wave * WPW + k) * 1024);
    ++d_idx;
    if constexpr (TAPS) {
      if (++d_cc == a.cpt) { d_cc = 0; ++d_tap; if (++d_tj == a.kw) { d_tj = 0; ++d_ti; } }
    }
    if (d_ok && ++d_k == g.ksteps) {
      d_k = 0;
      d_v += G;
      d_ok = decode(d_v, d_mt, d_nb);
      if constexpr (TAPS) { d_ti = d_tj = d_cc = d_tap = 0; item_rows(); }
    }
  };

  // ---- fragment addresses (ring slot 0; ks = 1 is ^ 64)
  const int frow = lane & 15, fq = lane >> 4;
  const int wrow0 = (frow >> 2) * 8 + (frow & 3);
  const int offW = ABYTES + (wn * 64 + wrow0) * 128 + ((fq ^ wkey(wrow0)) << 4);      // + (c>>1)*4096 + (c&1)*512
  int offA[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int r = wm * TP + p * 16 + frow;
    offA[p] = r * 128 + ((fq ^ (r & 7)) << 4);
  }

  f32x4 acc[NCT][NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int p = 0; p < NPT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  int c_v = blockIdx.x, c_idx = 0, c_mt, c_nb;
  if (!decode(c_v, c_mt, c_nb)) return;                // (uniform) nothing for this block

  // ---- prologue: NST - 1 K-steps in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue();
  int st_age = NST;                                    // K-steps since this wave's last epilogue stores (>= NST - 1: none pending)
  int st_cnt = 0;                                      // number of those stores

  // One K-step.  LAST = the item's last step: its epilogue inputs (forward: bias; data gradient: ReLU mask, old value) are asm loads
  // issued AHEAD of the step's DMA, waited for with vmcnt(DPW) after the MFMAs (only this step's DMA is younger) and handed to the
  // compiler by ONE statement naming every destination — straight-line code, so no register of a load in flight is ever copied.
  // Waves 0-3 issue the step's DMA BEFORE their fragment reads / MFMAs, waves 4-7 AFTER (one wave of each group per SIMD): a group's DMA
  // issue (~80 cycles per instruction) runs under the other group's MFMAs instead of all eight waves moving in lockstep.
  auto step = [&](auto lastc, auto latec) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(lastc)::value, LATE = decltype(latec)::value;
    constexpr int LDWAIT = LATE ? 0 : DPW;            // vector-memory operations younger than the epilogue loads at their wait
    // pieces of K-step c_idx have landed when at most the (NST - 2) younger steps + the stores of a recent epilogue are outstanding.
    // st_age = s - 1 at the start of the s-th step after an epilogue at step i.  Early waves issued DMA(i + NST - 1) BEFORE the stores of
    // step i, so the stores are younger than the awaited DMA(i + s) while s <= NST - 1; late waves issue that DMA AFTER their stores, so
    // for them the stores are younger only while s <= NST - 2 (tests/test_abi_cpu.py replays both issue orders against this rule).
    const int pend = st_age <= NST - 2 - (LATE ? 1 : 0) ? st_cnt : 0;
    pw_wait_vmcnt_dyn(pw_floor_count((NST - 2) * DPW + pend));
    __builtin_amdgcn_s_barrier();                      // ... for every wave; and every wave is done reading slot (c_idx - 1) % NST
    [[maybe_unused]] pw_u32x4 ld_m[NPT][NPAIR], ld_o[NPT][NPAIR], ld_b[NPAIR][2];
    const int cb = c_nb * BN + wn * 64 + fq * 8;       // this lane's 8 consecutive channels of pair 0 (+32 per pair)
    if constexpr (LAST && LD) {
      if constexpr (DGRAD) {
        // (a mode that is off reads the other mode's tensor: same lines, no extra traffic, value unused)
        const bf16_t* mbase = a.mask ? a.mask : reinterpret_cast<const bf16_t*>(a.y);
        const bf16_t* obase = a.accumulate ? reinterpret_cast<const bf16_t*>(a.y) : a.mask;
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          const long m = (long)c_mt * BM + wm * TP + p * 16 + frow;
          const long mm = m < a.M ? m : (long)a.M - 1;
#pragma unroll
          for (int q = 0; q < NPAIR; ++q) {
            // (a mode that is off reads the other mode's tensor at ITS pitch)
            ld_m[p][q] = pw_load16(mbase + (size_t)mm * (a.mask ? a.ldm : a.ldy) + cb + q * 32);
            ld_o[p][q] = pw_load16(obase + (size_t)mm * (a.accumulate ? a.ldy : a.ldm) + cb + q * 32);
          }
        }
      } else {
        const float* bbase = a.bias ? a.bias + cb : reinterpret_cast<const float*>(a.w);
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
          ld_b[q][0] = pw_load16(bbase + q * 32);
          ld_b[q][1] = pw_load16(bbase + q * 32 + 4);
        }
      }
    }
    if constexpr (!LATE) issue();                      // K-step c_idx + NST - 1 into the slot just freed
    const int base = (c_idx % NST) * SB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[NCT], xf[NPT];
#pragma unroll
      for (int c = 0; c < NCT; ++c) wf[c] = *reinterpret_cast<const bf16x8*>(smem + base + ((offW + (c >> 1) * 4096 + (c & 1) * 512) ^ (ks * 64)));
#pragma unroll
      for (int p = 0; p < NPT; ++p) xf[p] = *reinterpret_cast<const bf16x8*>(smem + base + (offA[p] ^ (ks * 64)));
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int p = 0; p < NPT; ++p) acc[c][p] = DH_MFMA_16x16x32(wf[c], xf[p], acc[c][p]);
    }
    ++c_idx;
    ++st_age;
    if constexpr (LAST) {
      // ---- epilogue: lane owns channels cb + q*32 .. +7 of pixel m
      if constexpr (!LD) {
      } else if constexpr (DGRAD) {
        if constexpr (NPT == 4)
          asm volatile("s_waitcnt vmcnt(%16)" : "+v"(ld_m[0][0]), "+v"(ld_m[0][1]), "+v"(ld_m[1][0]), "+v"(ld_m[1][1]), "+v"(ld_m[2][0]), "+v"(ld_m[2][1]),
                       "+v"(ld_m[3][0]), "+v"(ld_m[3][1]), "+v"(ld_o[0][0]), "+v"(ld_o[0][1]), "+v"(ld_o[1][0]), "+v"(ld_o[1][1]), "+v"(ld_o[2][0]),
                       "+v"(ld_o[2][1]), "+v"(ld_o[3][0]), "+v"(ld_o[3][1]) : "n"(LDWAIT) : "memory");
        else if constexpr (NPT == 2)
          asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ld_m[0][0]), "+v"(ld_m[0][1]), "+v"(ld_m[1][0]), "+v"(ld_m[1][1]), "+v"(ld_o[0][0]), "+v"(ld_o[0][1]),
                       "+v"(ld_o[1][0]), "+v"(ld_o[1][1]) : "n"(LDWAIT) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ld_m[0][0]), "+v"(ld_m[0][1]), "+v"(ld_o[0][0]), "+v"(ld_o[0][1]) : "n"(LDWAIT) : "memory");
        static_assert(NPT == 1 || NPT == 2 || NPT == 4, "one wait statement per wave-tile height");
      } else {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ld_b[0][0]), "+v"(ld_b[0][1]), "+v"(ld_b[1][0]), "+v"(ld_b[1][1]) : "n"(LDWAIT) : "memory");
      }
      float bv[NPAIR][8];
#pragma unroll
      for (int q = 0; q < NPAIR; ++q)
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[q][r] = 0.f;
      if constexpr (!DGRAD && LD) {
        if (a.bias) {
#pragma unroll
          for (int q = 0; q < NPAIR; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              bv[q][r] = __uint_as_float(ld_b[q][0][r]);
              bv[q][4 + r] = __uint_as_float(ld_b[q][1][r]);
            }
        }
      }
      int ns = 0;
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const long m = (long)c_mt * BM + wm * TP + p * 16 + frow;
        [[maybe_unused]] pw_u32x4 tq[2];
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][p][r]; v[4 + r] = acc[2 * q + 1][p][r]; }
          if constexpr (!DGRAD) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += bv[q][r];
            if (a.relu && cb + q * 32 < a.relu_co) {      // (relu_co is a multiple of 8: a lane's 8 channels are all on one side)
#pragma unroll
              for (int r = 0; r < 8; ++r) v[r] = dh_relu(v[r]);
            }
          } else {
            if (LD && a.mask) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float lo, hi;
                unpack2bf(ld_m[p][q][r], lo, hi);
                if (!(lo > 0.f)) v[2 * r] = 0.f;
                if (!(hi > 0.f)) v[2 * r + 1] = 0.f;
              }
            }
            if (LD && a.accumulate) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float lo, hi;
                unpack2bf(ld_o[p][q][r], lo, hi);
                v[2 * r] += lo;
                v[2 * r + 1] += hi;
              }
            }
          }
          const pw_u32x4 t = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          if constexpr (NPAIR == 2) {
            tq[q] = t;
          } else {
            if (m < a.M) *reinterpret_cast<pw_u32x4*>(reinterpret_cast<bf16_t*>(a.y) + (size_t)m * a.ldy + cb + q * 32) = t;
          }
          ++ns;
          acc[2 * q][p] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[2 * q + 1][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (NPAIR == 2) {
          // WHOLE 128-byte lines per store instruction (round 5; conv_c8.hip has the measurement): the wave's 64 channels are one line of a
          // pixel and a lane holds its [0, 64) and [64, 128) halves (q = 0, 1), so storing piece q from every lane wrote 16 half lines per
          // instruction (1.2-1.3 x the bytes in write traffic).  Neighbouring pixels (lane ^ 1: rows m, m ^ 1 - tiles start on even rows)
          // trade pieces: the even pixel's lane ends up with both low halves, the odd one with both high halves; instruction 1 writes the
          // even pixels' lines whole, instruction 2 the odd pixels'.  Same number of store instructions (the counted waits rely on it).
          const bool odd = frow & 1;
          pw_u32x4 give, got;
#pragma unroll
          for (int e = 0; e < 4; ++e) give[e] = odd ? tq[0][e] : tq[1][e];
#pragma unroll
          for (int e = 0; e < 4; ++e) got[e] = dh_lane_xor1(give[e]);
          const pw_u32x4 first = odd ? got : tq[0], second = odd ? tq[1] : got;
          const long me = m & ~1l;
          bf16_t* line = reinterpret_cast<bf16_t*>(a.y) + (size_t)me * a.ldy + cb + (odd ? 32 : 0);
          if (me < a.M) *reinterpret_cast<pw_u32x4*>(line) = first;
          if (me + 1 < a.M) *reinterpret_cast<pw_u32x4*>(line + a.ldy) = second;
        }
      }
      st_age = 0;
      st_cnt = ((long)c_mt * BM + BM <= (long)a.M) ? ns : 0;      // a ragged last tile may skip stores: never over-count what is pending
    }
    if constexpr (LATE) issue();
  };
  if (wave < 4) {
    for (;;) {
      for (int k = 0; k + 1 < g.ksteps; ++k) step(std::false_type{}, std::false_type{});
      step(std::true_type{}, std::false_type{});
      c_v += G;
      if (!decode(c_v, c_mt, c_nb)) break;
    }
  } else {
    for (;;) {
      for (int k = 0; k + 1 < g.ksteps; ++k) step(std::false_type{}, std::true_type{});
      step(std::true_type{}, std::true_type{});
      c_v += G;
      if (!decode(c_v, c_mt, c_nb)) break;
    }
  }
  pw_wait_vmcnt<0>();                                  // zero-fill pieces of the steps beyond the stream may still be landing
}

template <int BN, int NST, bool DGRAD, bool LD, bool TAPS>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_pointwise_kernel(const ConvArgs a, const PwGeom g) {
  pw_body<BN, NST, DGRAD, LD, TAPS, false>(a, g);
}
template <int BN, int NST>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_pointwise_concat2_kernel(const ConvArgs a, const PwGeom g) {
  pw_body<BN, NST, false, true, false, true>(a, g);
}

int pw_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

template <int BN, int NST, bool DGRAD, bool LD, bool TAPS>
int launch_pw_t(const ConvArgs& a, hipStream_t s) {
  constexpr int LDS = NST * (128 * 128 + BN * 128);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pointwise_kernel<BN, NST, DGRAD, LD, TAPS>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  PwGeom g{};
  g.m_tiles = (a.M + 127) / 128;
  g.NB = a.Co / BN;
  g.items = g.m_tiles * g.NB;
  g.ksteps = a.Kpad / 64;
  g.div_nb = make_fastdiv(g.NB);
  int G = pw_cu_count();
  if (g.items < G) G = g.items;
  g.grouped = (G % 8 == 0 && g.items >= G) ? 1 : 0;
  g.per_xcd = (g.items + 7) / 8;
  hipLaunchKernelGGL((conv_pointwise_kernel<BN, NST, DGRAD, LD, TAPS>), dim3(G), dim3(512), LDS, s, a, g);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

template <int BN, int NST>
int launch_pw_concat2(const ConvArgs& a, hipStream_t s) {
  constexpr int LDS = NST * (128 * 128 + BN * 128);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pointwise_concat2_kernel<BN, NST>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  PwGeom g{};
  g.m_tiles = (a.M + 127) / 128;
  g.NB = a.Co / BN;
  g.items = g.m_tiles * g.NB;
  g.ksteps = a.Kpad / 64;
  g.div_nb = make_fastdiv(g.NB);
  int G = pw_cu_count();
  if (g.items < G) G = g.items;
  g.grouped = (G % 8 == 0 && g.items >= G) ? 1 : 0;
  g.per_xcd = (g.items + 7) / 8;
  hipLaunchKernelGGL((conv_pointwise_concat2_kernel<BN, NST>), dim3(G), dim3(512), LDS, s, a, g);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

template <int BN, int NST, bool DGRAD, bool LD>
int launch_pw(const ConvArgs& a, hipStream_t s) {
  const bool plain = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.H == a.Ho && a.W == a.Wo;
  return plain ? launch_pw_t<BN, NST, DGRAD, LD, false>(a, s) : launch_pw_t<BN, NST, DGRAD, LD, true>(a, s);
}

bool pw_eligible(const ConvArgs& a) {
  if (a.dstride != 1 || a.taps > 32) return false;
  if (a.C % 64 != 0 || a.Co % 64 != 0 || a.C < 64 || a.Kpad != a.taps * a.C) return false;
  if (a.out_f32 || a.resid || a.pool_y) return false;
  if ((long)a.M < 2048) return false;                      // tiny maps: the flat-M kernel's smaller tiles fill the chip better
  if ((long)a.N * a.H * a.W * a.ldx >= (1l << 31) || (long)a.Co * a.Kpad >= (1l << 31)) return false;
  if ((a.ldx | a.ldy | a.ldm | a.relu_co) & 7) return false;         // 16-byte pieces: pitches (and the ReLU boundary) are multiples of 8 channels
  return true;
}

}  // namespace

// tile width: the whole Co where it fits (X read once).  A data gradient WITH epilogue inputs holds them in registers across its last
// K-step: 128 wide by default; option pw_dgrad_ld_bn = 256 lets it take the 256-wide tile too (32 more registers; measured no faster)
static int pw_bn(const ConvArgs& a, bool dgrad_ld) {
  const int cap = dgrad_ld ? danhip_option("pw_dgrad_ld_bn") : 256;
  return (a.Co % 256 == 0 && cap >= 256) ? 256 : (a.Co % 128 == 0 ? 128 : 64);
}

const char* danhip_conv_pointwise_label(const ConvArgs& a, bool dgrad) {
  if (!pw_eligible(a)) return nullptr;
  const bool ld = dgrad ? (a.mask || a.accumulate) : true;
  const int bn = pw_bn(a, dgrad && ld);
  const bool plain = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.H == a.Ho && a.W == a.Wo;
  static const char* names[2][2][2][3] = {      // [taps][dgrad][ld][bn 256 / 128 / 64]
      {{{"conv_pointwise_kernel<256, 3, false, false, false>", "conv_pointwise_kernel<128, 4, false, false, false>", "conv_pointwise_kernel<64, 4, false, false, false>"},
        {"conv_pointwise_kernel<256, 3, false, true, false>", "conv_pointwise_kernel<128, 4, false, true, false>", "conv_pointwise_kernel<64, 4, false, true, false>"}},
       {{"conv_pointwise_kernel<256, 3, true, false, false>", "conv_pointwise_kernel<128, 4, true, false, false>", "conv_pointwise_kernel<64, 4, true, false, false>"},
        {"conv_pointwise_kernel<256, 3, true, true, false>", "conv_pointwise_kernel<128, 4, true, true, false>", "conv_pointwise_kernel<64, 4, true, true, false>"}}},
      {{{"conv_pointwise_kernel<256, 3, false, false, true>", "conv_pointwise_kernel<128, 4, false, false, true>", "conv_pointwise_kernel<64, 4, false, false, true>"},
        {"conv_pointwise_kernel<256, 3, false, true, true>", "conv_pointwise_kernel<128, 4, false, true, true>", "conv_pointwise_kernel<64, 4, false, true, true>"}},
       {{"conv_pointwise_kernel<256, 3, true, false, true>", "conv_pointwise_kernel<128, 4, true, false, true>", "conv_pointwise_kernel<64, 4, true, false, true>"},
        {"conv_pointwise_kernel<256, 3, true, true, true>", "conv_pointwise_kernel<128, 4, true, true, true>", "conv_pointwise_kernel<64, 4, true, true, true>"}}}};
  return names[plain ? 0 : 1][dgrad ? 1 : 0][ld ? 1 : 0][bn == 256 ? 0 : bn == 128 ? 1 : 2];
}

// DANHIP_OK when launched, 1 when the shape is not eligible (caller falls back to the flat-M kernel).
int danhip_launch_conv_pointwise(const ConvArgs& a, hipStream_t s) {
  if (!pw_eligible(a)) return 1;
  if (a.x2) {                                          // forward over the concatenation of two sources (danhip_conv2d_fwd_concat2)
    const bool plain = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.H == a.Ho && a.W == a.Wo;
    if (!plain || a.mask || a.accumulate || a.ksplit <= 0 || a.ksplit >= a.Kpad / 64) return 1;
    const int bn2 = pw_bn(a, false);
    return bn2 == 256 ? launch_pw_concat2<256, 3>(a, s) : bn2 == 128 ? launch_pw_concat2<128, 4>(a, s) : launch_pw_concat2<64, 4>(a, s);
  }
  const bool dgrad = !a.bias && !a.relu;
  if (!dgrad && (a.mask || a.accumulate)) return 1;
  const bool ld = dgrad ? (a.mask || a.accumulate) : true;
  const int bn = pw_bn(a, dgrad && ld);
  if (dgrad) {
    if (ld) return bn == 256 ? launch_pw<256, 3, true, true>(a, s) : bn == 128 ? launch_pw<128, 4, true, true>(a, s) : launch_pw<64, 4, true, true>(a, s);
    return bn == 256 ? launch_pw<256, 3, true, false>(a, s) : bn == 128 ? launch_pw<128, 4, true, false>(a, s) : launch_pw<64, 4, true, false>(a, s);
  }
  return bn == 256 ? launch_pw<256, 3, false, true>(a, s) : bn == 128 ? launch_pw<128, 4, false, true>(a, s) : launch_pw<64, 4, false, true>(a, s);
}
