// Halo-reuse 3x3 / stride-1 convolution on MFMA for gfx950 (bf16 NHWC in, fp32 accumulate) — the kernel that carries the
// backbone (net/sfd_net.py:127-156 conv1_2 .. conv4_3), LFPN fused convs (net/pb_net.py:185-226, net/danet.py:339-380),
// shared head convs (net/danet.py:469-532), the PyramidBox CPM (net/pb_net.py:158-183) and, run on dY with tap-flipped
// weights, their data gradients.
//
// One PERSISTENT 512-thread workgroup per CU walks work items (spatial tile TH x TW of one image) x (BN output channels).
// For every 64-channel chunk of the input the (TH+2) x (TW+2) halo patch is DMA'd into LDS ONCE ([pixel][64 ch] = 128-byte
// rows, 16-byte XOR swizzle on the source side) and all nine taps read it at shifted row offsets: the activation is
// fetched once instead of nine times (the flat-M kernel re-gathers it per tap).  Weight tiles [BN][64] (one per tap and
// chunk) stream through an NSW-deep LDS ring.  All DMAs are global_load_lds_dwordx4; the loop never drains them:
// a counted s_waitcnt vmcnt(N) + raw s_barrier per step keeps NSW-1 weight tiles and the next patch in flight.
// The step sequence is flattened across chunks and items, so the next item's first patch and weights are already
// landing while the current item's epilogue runs.
//
// MFMA: v_mfma_f32_16x16x32_bf16, weight fragment = A operand, pixel fragment = B operand -> a lane owns 4 consecutive
// output channels of one pixel (8-byte bf16 / 16-byte fp32 stores into NHWC).
#include <type_traits>

#include "conv_common.h"

namespace {

struct HaloGeom {
  int tiles_x, tiles_y;     // spatial tiles per image
  int sp_items;             // N * tiles_y * tiles_x
  int NB;                   // output-channel blocks (Co / BN)
  int cch;                  // 64-channel chunks of the input (C / 64)
  int grouped;              // 1: XCD-grouped item mapping (the NB blocks of one spatial tile run on one XCD)
  FastDiv div_tx, div_txy, div_nb;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// Epilogue of the halo kernel for one lane's 4 consecutive channels (Co % 64 == 0: always a full, aligned quad).
__device__ __forceinline__ void halo_store4(const ConvArgs& a, const f32x4& acc, size_t m, int co) {
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const size_t o = m * (size_t)a.Co + co;
  if (a.bias) {
    const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (a.relu) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (a.out_f32) {
    float* y = reinterpret_cast<float*>(a.y) + o;
    float4 t = make_float4(v[0], v[1], v[2], v[3]);
    if (a.accumulate) { const float4 u = *reinterpret_cast<const float4*>(y); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(y) = t;
    return;
  }
  bf16_t* y = reinterpret_cast<bf16_t*>(a.y) + o;
  if (a.mask) {
    const uint2 mk = *reinterpret_cast<const uint2*>(a.mask + o);
    const bf16_t* mp = reinterpret_cast<const bf16_t*>(&mk);
#pragma unroll
    for (int r = 0; r < 4; ++r) if (!(bf2f(mp[r]) > 0.f)) v[r] = 0.f;
  }
  if (a.resid) {
    const uint2 rs = *reinterpret_cast<const uint2*>(a.resid + o);
    const bf16_t* rp = reinterpret_cast<const bf16_t*>(&rs);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bf2f(rp[r]);
  }
  if (a.accumulate) {
    const uint2 old = *reinterpret_cast<const uint2*>(y);
    const bf16_t* op = reinterpret_cast<const bf16_t*>(&old);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bf2f(op[r]);
  }
  uint2 t;
  t.x = pack2bf(v[0], v[1]);
  t.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(y) = t;
}

// Zeros for out-of-image halo pixels: LDS-DMA lanes keep a per-lane 64-bit source pointer that is bumped by 128 bytes
// per 64-channel chunk, so the page must cover C*2 + 16 bytes (C <= 2048).
__device__ __attribute__((aligned(64))) unsigned g_halo_zero[1056] = {0};

__device__ __forceinline__ void glds16_so(const void* sbase, unsigned voff, void* lds_wave_base) {   // saddr + 32-bit voffset
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(reinterpret_cast<const char*>(sbase) + voff), (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

template <int TH, int TW, int BN, int WM, int WN, int NSW>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo_kernel(const ConvArgs a, const HaloGeom g) {
  constexpr int PW = TW + 2;                       // patch row pitch (pixels); even, so LDS row parity == column parity
  constexpr int PROWS = (TH + 2) * PW;             // patch pixels
  constexpr int PPIECES = (PROWS + 7) / 8;         // 1 KiB DMA pieces per patch (8 pixel rows each)
  constexpr int PBYTES = PPIECES * 1024;
  constexpr int PL = (PPIECES + 7) / 8;            // patch pieces per wave
  constexpr int WBYTES = BN * 128;                 // one weight tile
  constexpr int WL = BN / 64;                      // weight pieces per wave per step
  constexpr int D = NSW - 1;                       // weight prefetch distance (steps)
  constexpr int BM = TH * TW;
  constexpr int TP = BM / WM, TC = BN / WN;        // wave tile: pixels x channels
  constexpr int NPT = TP / 16, NCT = TC / 16;
  static_assert(WM * WN == 8, "8 waves");
  static_assert(TP % 16 == 0 && TC % 16 == 0 && TW % 16 == 0, "MFMA tile alignment");
  static_assert(BN % 64 == 0 && NSW == 4 && (PW % 2) == 0, "layout assumptions");
  static_assert(2 * PW * 128 + 64 + PBYTES < 65536, "ds_read immediate offsets");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS map: [patch 0][patch 1][weight ring: NSW tiles]
  constexpr int WRING = 2 * PBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int srow = lane >> 3;

  // ---- item mapping --------------------------------------------------------------------------------------------
  const int G = gridDim.x;
  auto decode = [&](int v, int& sp, int& nb) -> bool {     // v = round * G + block
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
  auto sp_coords = [&](int sp, int& n, int& y0, int& x0) {
    n = (int)fdiv((unsigned)sp, g.div_txy);
    const int rem = sp - n * (g.tiles_x * g.tiles_y);
    const int ty = (int)fdiv((unsigned)rem, g.div_tx);
    y0 = ty * TH;
    x0 = (rem - ty * g.tiles_x) * TW;
  };

  // ---- patch DMA: per-lane geometry (tile independent) and per-item source pointers ------------------------------------
  // LDS row R = hy*PW + hx holds pixel (y0-1+hy, x0-1+hx); the 16-byte chunk c of its 64 channels sits at position
  // c ^ (hx & 7) (swizzle keyed on the patch COLUMN: a tap's row shift then is a pure address offset on the read side).
  int pgeo[PL];                                    // (hy << 8) | hx, or -1 when the slot is beyond the patch
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    int piece = k * 8 + wave;
    if (piece > PPIECES - 1) piece = PPIECES - 1;  // duplicate load of the last piece keeps the per-wave DMA count uniform
    const int row = piece * 8 + srow;
    const int hy = row / PW, hx = row - hy * PW;
    pgeo[k] = row < PROWS ? ((hy << 8) | hx) : -1;
  }
  int p_v = blockIdx.x, p_cc = 0, p_idx = 0;       // patch cursor: next chunk to load; p_idx selects the buffer
  int p_sp, p_nb;
  bool p_ok = decode(p_v, p_sp, p_nb);
  const char* psrc[PL];                            // this lane's source pointer per piece for chunk p_cc
  auto patch_item_setup = [&]() {
    int n, y0, x0;
    sp_coords(p_sp, n, y0, x0);
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int hx = pgeo[k] & 255;
      const int y = y0 - 1 + (pgeo[k] >> 8), x = x0 - 1 + hx;
      const bool ok = pgeo[k] >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const unsigned off = (unsigned)(((n * a.H + y) * a.W + x) * a.C) * 2u + (unsigned)(((lane & 7) ^ (hx & 7)) << 4);
      psrc[k] = ok ? reinterpret_cast<const char*>(a.x) + off : reinterpret_cast<const char*>(g_halo_zero);
    }
  };
  auto issue_patch = [&]() {                       // loads chunk (p_v, p_cc) into buffer p_idx & 1 and advances the cursor
    char* dst = smem + (p_idx & 1) * PBYTES;
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      int piece = k * 8 + wave;
      if (piece > PPIECES - 1) piece = PPIECES - 1;
      glds16(psrc[k], dst + piece * 1024);
      psrc[k] += 128;
    }
    ++p_idx;
    if (++p_cc == g.cch) {
      p_cc = 0;
      p_v += G;
      p_ok = decode(p_v, p_sp, p_nb);
      if (p_ok) patch_item_setup();
    }
  };
  if (p_ok) patch_item_setup();

  // ---- weight DMA cursor: step (w_v, w_cc, w_tap) to be loaded next ------------------------------------------------------
  int w_v = blockIdx.x, w_cc = 0, w_tap = 0, w_idx = 0;
  int w_sp, w_nb;
  bool w_ok = decode(w_v, w_sp, w_nb);
  const unsigned wlane = (unsigned)((wave * WL * 8 + srow) * a.Kpad) * 2u + (unsigned)(((lane & 7) ^ srow) << 4);   // + k*8 rows
  auto issue_w = [&]() {
    char* dst = smem + WRING + (w_idx & (NSW - 1)) * WBYTES;
    const char* sbase = reinterpret_cast<const char*>(a.w) + ((size_t)(w_nb * BN) * a.Kpad + w_tap * a.C + w_cc * 64) * 2;   // uniform
#pragma unroll
    for (int k = 0; k < WL; ++k) glds16_so(sbase, wlane + (unsigned)(k * 8 * a.Kpad) * 2u, dst + (wave * WL + k) * 1024);
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

  // ---- compute-side per-lane constants --------------------------------------------------------------------------
  const int frow = lane & 15, fq = lane >> 4;
  const int offW = WRING + (wn * TC + frow) * 128 + ((fq ^ (frow & 7)) << 4);     // + stage*WBYTES, + c*2048, ^ ks*64
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

  // Fragment loads of k-slice ks of the step at (weight stage base wbase, patch buffer offset pofs, tap TAP).
  auto load_frags = [&](bf16x8 (&wf)[NCT], bf16x8 (&xf)[NPT], int wbase, int pofs, auto tapc, int ks) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr int TI = TAP / 3, TJ = TAP % 3;
    const int wb = ks ? (wbase ^ 64) : wbase;
#pragma unroll
    for (int c = 0; c < NCT; ++c) wf[c] = *reinterpret_cast<const bf16x8*>(smem + wb + c * 2048);
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const int pa = (ks ? (pxaddr[TJ][p] ^ 64) : pxaddr[TJ][p]) + pofs;
      xf[p] = *reinterpret_cast<const bf16x8*>(smem + pa + TI * PW * 128);
    }
  };
  auto mma = [&](const bf16x8 (&wf)[NCT], const bf16x8 (&xf)[NPT], auto c0c, auto c1c) {      // channel tiles [C0, C1)
#pragma unroll
    for (int c = decltype(c0c)::value; c < decltype(c1c)::value; ++c)
#pragma unroll
      for (int p = 0; p < NPT; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xf[p], acc[c][p], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using IN = std::integral_constant<int, NCT>;

  // ---- prologue -----------------------------------------------------------------------------------------------------
  int c_v = blockIdx.x, c_sp, c_nb;
  bool c_ok = decode(c_v, c_sp, c_nb);
  if (!c_ok) return;                               // (uniform) nothing to do for this block
  issue_patch();
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (w_ok) issue_w();
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // Software pipeline over half-steps: while the 16 MFMAs of one 32-deep k-slice run, the fragments of the next slice are
  // read from LDS — including the first slice of the NEXT step, whose weight tile became visible one step earlier
  // (the hand-off at the end of a step waits for the tile of step + 2).  One step = one tap of one 64-channel chunk.
  int chunk = 0, cc = 0;                           // running chunk number (patch buffer = chunk & 1; stage = (chunk + tap) & 3)
  bf16x8 wf0[NCT], xf0[NPT], wf1[NCT], xf1[NPT];
  load_frags(wf0, xf0, offW, 0, std::integral_constant<int, 0>{}, 0);
  for (;;) {
    const int pofs = (chunk & 1) * PBYTES;
    bool patch_issued = false;
    auto step_body = [&](auto tapc) {
      constexpr int TAP = decltype(tapc)::value;
      // -- prefetch: weights of step + D, and (at tap 0) the next chunk's patch
      const bool more_w = w_ok;
      if (more_w) issue_w();
      if (TAP == 0 && p_ok) { issue_patch(); patch_issued = true; }
      const int wbase = offW + ((chunk + TAP) & (NSW - 1)) * WBYTES;
      // -- slice 1 fragment reads, then slice 0 MFMAs.  The slice-0 fragments were read one half-step (and a barrier)
      //    ago: retire them explicitly so the compiler does not make the MFMAs below wait for the reads issued here.
      __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
      load_frags(wf1, xf1, wbase, pofs, tapc, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(wf0, xf0, I0{}, IN{});
      __builtin_amdgcn_sched_barrier(0);
      // -- slice 1 MFMAs; the next step's slice 0 fragment reads go in after the first channel tile so that the wait in
      //    front of the first slice-1 MFMA covers only the (long finished) slice-1 reads
      constexpr int NTAP = (TAP + 1) % 9;
      const int nwbase = offW + ((chunk + TAP + 1) & (NSW - 1)) * WBYTES;
      const int npofs = TAP == 8 ? (PBYTES - pofs) : pofs;
      mma(wf1, xf1, I0{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
      load_frags(wf0, xf0, nwbase, npofs, std::integral_constant<int, NTAP>{}, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(wf1, xf1, I1{}, IN{});
      __builtin_amdgcn_sched_barrier(0);
      // -- hand-off: the weight tile of step + 2 (and any patch issued before it) must have landed
      if (!more_w) {
        wait_vmcnt<0>();                           // tail of this block's work
      } else if (TAP < 2 && patch_issued) {
        wait_vmcnt<(D - 2) * WL + PL>();
      } else {
        wait_vmcnt<(D - 2) * WL>();
      }
      __builtin_amdgcn_s_barrier();
    };
    step_body(std::integral_constant<int, 0>{});
    step_body(std::integral_constant<int, 1>{});
    step_body(std::integral_constant<int, 2>{});
    step_body(std::integral_constant<int, 3>{});
    step_body(std::integral_constant<int, 4>{});
    step_body(std::integral_constant<int, 5>{});
    step_body(std::integral_constant<int, 6>{});
    step_body(std::integral_constant<int, 7>{});
    step_body(std::integral_constant<int, 8>{});
    ++chunk;
    if (++cc == g.cch) {
      // ---- epilogue of this item ---------------------------------------------------------------------------------
      cc = 0;
      int n, y0, x0;
      sp_coords(c_sp, n, y0, x0);
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const int t = wm * TP + p * 16 + frow;
        const int y = y0 + t / TW, x = x0 + t % TW;
        const bool ok = y < a.H && x < a.W;
        const size_t m = (size_t)((n * a.H + y) * a.W + x);
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          if (ok) halo_store4(a, acc[c][p], m, c_nb * BN + wn * TC + c * 16 + fq * 4);
          acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      c_v += G;
      c_ok = decode(c_v, c_sp, c_nb);
      if (!c_ok) break;
    }
  }
}

struct HaloPlan {
  int th, tw, bn;
};

// Picks the tile; returns false when the shape should go to the flat-M kernel.
bool plan_halo(const ConvArgs& a, HaloPlan* p) {
  if (!(a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dstride == 1 && a.pad_t == 1 && a.pad_l == 1)) return false;
  if (a.H != a.Ho || a.W != a.Wo) return false;
  if (a.C % 64 != 0 || a.Co % 64 != 0 || a.C > 2048) return false;
  if ((int64_t)a.Co * a.Kpad >= (1ll << 31)) return false;
  auto util = [&](int th, int tw) {
    const double ph = (double)((a.H + th - 1) / th * th), pw = (double)((a.W + tw - 1) / tw * tw);
    return (double)a.H * a.W / (ph * pw);
  };
  const double u1 = util(8, 32), u2 = util(16, 16);
  if (u1 >= u2) { p->th = 8; p->tw = 32; } else { p->th = 16; p->tw = 16; }
  if ((u1 > u2 ? u1 : u2) < 0.78) return false;
  p->bn = (a.Co % 128 == 0) ? 128 : 64;
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

template <int TH, int TW, int BN, int WM, int WN, int NSW>
int launch_halo_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int PPIECES = ((TH + 2) * (TW + 2) + 7) / 8;
  constexpr int LDS = 2 * PPIECES * 1024 + NSW * BN * 128;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<TH, TW, BN, WM, WN, NSW>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  HaloGeom g{};
  g.tiles_x = (a.W + TW - 1) / TW;
  g.tiles_y = (a.H + TH - 1) / TH;
  g.sp_items = a.N * g.tiles_x * g.tiles_y;
  g.NB = a.Co / BN;
  g.cch = a.C / 64;
  g.div_tx = make_fastdiv(g.tiles_x);
  g.div_txy = make_fastdiv(g.tiles_x * g.tiles_y);
  g.div_nb = make_fastdiv(g.NB);
  const long items = (long)g.sp_items * g.NB;
  int G = cu_count();
  g.grouped = 0;
  if (items >= G && (G % 8) == 0 && ((G / 8) % g.NB) == 0 && g.NB > 1) g.grouped = 1;
  if (items < G) G = (int)items;
  hipLaunchKernelGGL((conv3x3_halo_kernel<TH, TW, BN, WM, WN, NSW>), dim3(G), dim3(512), LDS, s, a, g);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

}  // namespace

int danhip_launch_conv_halo(const ConvArgs& a, hipStream_t s) {
  HaloPlan p;
  if (!plan_halo(a, &p)) return 1;
  if (p.th == 8) {
    if (p.bn == 128) return launch_halo_cfg<8, 32, 128, 4, 2, 4>(a, s);
    return launch_halo_cfg<8, 32, 64, 8, 1, 4>(a, s);
  }
  if (p.bn == 128) return launch_halo_cfg<16, 16, 128, 4, 2, 4>(a, s);
  return launch_halo_cfg<16, 16, 64, 8, 1, 4>(a, s);
}

const char* danhip_conv_halo_label(const ConvArgs& a) {
  HaloPlan p;
  if (!plan_halo(a, &p)) return nullptr;
  if (p.th == 8) return p.bn == 128 ? "conv3x3_halo_kernel<8, 32, 128, 4, 2, 4>" : "conv3x3_halo_kernel<8, 32, 64, 8, 1, 4>";
  return p.bn == 128 ? "conv3x3_halo_kernel<16, 16, 128, 4, 2, 4>" : "conv3x3_halo_kernel<16, 16, 64, 8, 1, 4>";
}
