// Weight gradient of the NHWC convolution on MFMA (gfx950): dW[tap][ci][co] += sum_m X[m@tap][ci] * dY[m][co].
//
// GEMM view per tap: C[ci, co] = A^T[ci, m] * B[m, co] with the reduction over PIXELS, which is the slow (strided)
// axis of both NHWC operands.  Both tiles are therefore staged exactly as they lie in HBM ([pixel][channel] rows,
// 16-byte LDS-DMA pieces) and the MFMA fragments are produced by the gfx950 transpose read ds_read_b64_tr_b16
// (a 4-pixel x 16-channel block per 16-lane group), so nothing is ever transposed in registers or through HBM.
// The reduction over pixels is split across workgroups (grid.z); partial tiles are combined with fp32 atomics
// straight into the HWIO gradient (zeroed by the caller once per step).
#include <cstdlib>

#include "conv_common.h"

namespace {

struct WgradArgs {
  const bf16_t* x;     // [N,H,W,C]
  const bf16_t* dy;    // [N,Ho,Wo,Co8]
  float* dw;           // [kh,kw,cin_real,Cout]
  float* db;           // [Cout] bias gradient (column sums of dy) or null
  int N, H, W, C, Ho, Wo, Co8, Cout, cin_real;
  int ldx, ldy;        // pixel pitches of x / dy in elements (channel-slice views; dense: C / Co8)
  int kh, kw, stride, pad_t, pad_l;
  int M, ktiles, kt_per_split;
  int linear;
  int ci_tiles;        // number of input-channel tiles (blockIdx.x = co_tile * ci_tiles + ci_tile)
  int tapcols;         // 1: tile columns are the taps (C == 8 first layer): column chunk t = tap t, channels 0..7
  FastDiv div_wo, div_howo;
};

template <int RB>
__device__ __forceinline__ int swz(int pix) {  // 32-byte-slot XOR making the transpose reads conflict-free
  if (RB == 256) return ((pix & 3) | (((pix >> 3) & 1) << 2)) << 1;
  if (RB == 128) return (((pix >> 1) & 1) | (((pix >> 3) & 1) << 1)) << 1;
  return 0;
}

template <int BCI, int BCO, int WCI_WAVES>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int RBX = BCI * 2, RBY = BCO * 2;            // row bytes of the two tiles
  constexpr int XB = 64 * RBX, YB = 64 * RBY;            // tile bytes (64 pixels per K tile)
  constexpr int STAGE = XB + YB;
  constexpr int PX = XB / 1024, PY = YB / 1024;          // 1 KiB LDS-DMA pieces per tile
  constexpr int RPX = 1024 / RBX, RPY = 1024 / RBY;      // pixel rows per piece
  constexpr int WCO_WAVES = 4 / WCI_WAVES;
  constexpr int TCI = BCI / WCI_WAVES, TCO = BCO / WCO_WAVES;
  constexpr int NI = TCI / 16, NO = TCO / 16;
  static_assert(TCI % 16 == 0 && TCO % 16 == 0, "wave tile must be MFMA aligned");
  constexpr int RX = (PX + 3) / 4, RY = (PY + 3) / 4;    // DMA rounds per wave

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave / WCO_WAVES, wco = wave % WCO_WAVES;

  const int ci_tile = blockIdx.x % a.ci_tiles, co_tile = blockIdx.x / a.ci_tiles;
  const int ci0 = ci_tile * BCI, co0 = co_tile * BCO;
  const int tap = a.tapcols ? 0 : blockIdx.y;
  const int ti = tap / a.kw, tj = tap % a.kw;
  const int kt_begin = blockIdx.z * a.kt_per_split;
  const int kt_end = min(a.ktiles, kt_begin + a.kt_per_split);
  if (kt_begin >= kt_end) return;

  // per-lane staging geometry
  const int xr = (lane * 16) / RBX, xc = ((lane * 16) % RBX) / 16;   // row in piece, chunk position
  const int yr = (lane * 16) / RBY, yc = ((lane * 16) % RBY) / 16;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_danhip_zero_page);

  auto stage = [&](int kt, int buf) {
    char* sX = smem + buf * STAGE;
    char* sY = sX + XB;
    const int mb = kt * 64;
#pragma unroll
    for (int r = 0; r < RX; ++r) {
      const int piece = r * 4 + wave;
      if (PX % 4 == 0 || piece < PX) {
        const int prow = piece * RPX + xr;
        const int m = mb + prow;
        const int cs = xc ^ swz<RBX>(prow);                           // source chunk for this LDS position
        const bf16_t* src = zero;
        if (a.linear) {                                               // pointwise, stride 1: pixel m of x IS pixel m of dy, no decomposition
          const int cc = ci0 + cs * 8;
          if (m < a.M && cc < a.C) src = a.x + ((size_t)m * a.ldx + cc);
        } else if (m < a.M) {
          const unsigned n = fdiv((unsigned)m, a.div_howo);
          const unsigned rem = (unsigned)m - n * (unsigned)(a.Ho * a.Wo);
          const unsigned ho = fdiv(rem, a.div_wo);
          const unsigned wo = rem - ho * (unsigned)a.Wo;
          int i = ti, j = tj, cc = ci0 + cs * 8;
          bool ok = true;
          if (a.tapcols) { i = cs / a.kw; j = cs % a.kw; cc = 0; ok = cs < a.kh * a.kw; }
          const int hi = (int)ho * a.stride - a.pad_t + i, wi = (int)wo * a.stride - a.pad_l + j;
          ok = ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W && cc < a.C;
          if (ok) src = a.x + ((size_t)(((int)n * a.H + hi) * a.W + wi) * a.ldx + cc);
        }
        glds16(src, sX + piece * 1024);
      }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      const int piece = r * 4 + wave;
      if (PY % 4 == 0 || piece < PY) {
        const int prow = piece * RPY + yr;
        const int m = mb + prow;
        const int cs = yc ^ swz<RBY>(prow);
        const int cc = co0 + cs * 8;
        const bf16_t* src = (m < a.M && cc < a.Co8) ? a.dy + ((size_t)m * a.ldy + cc) : zero;
        glds16(src, sY + piece * 1024);
      }
    }
  };

  f32x4 acc[NI][NO];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[i][o] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;

  // Bias gradient db[co] = sum_m dy[m][co] rides along as one extra MFMA row of ones (no extra pass over dy): done by
  // the wci == 0 waves of the blocks that own tap 0 / input-channel tile 0.
  const bool do_bias = a.db != nullptr && blockIdx.y == 0 && ci_tile == 0 && wci == 0;
  f32x4 accb[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (act16_t)1.0f;

  stage(kt_begin, 0);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < kt_end) stage(kt + 1, buf ^ 1);
    const char* sX = smem + buf * STAGE;
    const char* sY = sX + XB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x4 xh[2][NI], yh[2][NO];   // halves of the fragments: k = 8g+0..3 and 8g+4..7
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pix = ks * 32 + 8 * g + 4 * h + q;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int cb = (wci * TCI + i * 16) / 16;
          const int ch = cb * 2 + (p >> 1);
          xh[h][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (LDS_AS s16x4*)(sX + pix * RBX + ((ch ^ swz<RBX>(pix)) << 4) + (p & 1) * 8));
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          const int cb = (wco * TCO + o * 16) / 16;
          const int ch = cb * 2 + (p >> 1);
          yh[h][o] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (LDS_AS s16x4*)(sY + pix * RBY + ((ch ^ swz<RBY>(pix)) << 4) + (p & 1) * 8));
        }
      }
      // whole-vector concatenation + bitcast (element-wise extraction of the tr-read result miscompiles on ROCm 7.2:
      // hipcc splats element 0)
      bf16x8 xf[NI], yf[NO];
#pragma unroll
      for (int i = 0; i < NI; ++i)
        xf[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xh[0][i], xh[1][i], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int o = 0; o < NO; ++o)
        yf[o] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(yh[0][o], yh[1][o], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[i][o] = DH_MFMA_16x16x32(xf[i], yf[o], acc[i][o]);
      if (do_bias) {
#pragma unroll
        for (int o = 0; o < NO; ++o) accb[o] = DH_MFMA_16x16x32(ones, yf[o], accb[o]);
      }
    }
  }
  if (do_bias && g == 0) {                                // every row of accb holds the column sums; take row 0
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const int co = co0 + wco * TCO + o * 16 + (lane & 15);
      if (co < a.Cout) atomicAdd(a.db + co, accb[o][0]);
    }
  }

  // epilogue: lane holds C[ci = i*16 + (lane>>4)*4 + r][co = o*16 + (lane&15)]
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cil = wci * TCI + i * 16 + g * 4 + r;                 // column inside the X tile
      int t = tap, ci = ci0 + cil;
      if (a.tapcols) { t = cil / 8; ci = cil % 8; if (t >= a.kh * a.kw) continue; }
      if (ci >= a.cin_real) continue;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const int co = co0 + wco * TCO + o * 16 + (lane & 15);
        if (co < a.Cout) atomicAdd(a.dw + ((size_t)(t * a.cin_real + ci) * a.Cout + co), acc[i][o][r]);
      }
    }
  }
}

template <int BCI, int BCO, int WCI_WAVES>
int launch_wgrad(WgradArgs& a, hipStream_t s) {
  static const bool attr_ok = (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<BCI, BCO, WCI_WAVES>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (BCI + BCO) * 2) == hipSuccess);
  (void)attr_ok;
  const int co_tiles = cdiv(a.Co8, BCO);
  a.ci_tiles = a.tapcols ? 1 : cdiv(a.C, BCI);
  const int taps = a.tapcols ? 1 : a.kh * a.kw;
  const int base_blocks = co_tiles * a.ci_tiles * taps;
  // split the pixel reduction so that every CU holds as many workgroups as fit (LDS: two stages of the X and dY tiles; at most 4) in ONE
  // wave of the grid — 1024 blocks of the 48 KB <128, 64> instance were 1.33 waves of 768 slots: the first layer's gradient ran its last
  // third alone on a third of the chip — but keep >= 8 K tiles per split
  const int lds_block = 2 * 64 * (BCI + BCO) * 2;
  int per_cu = (160 * 1024) / lds_block;
  if (per_cu > 4) per_cu = 4;
  static const int cus = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  const int slots = per_cu * cus;
  int splits = slots / base_blocks;
  const int max_splits = (a.ktiles + 7) / 8;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.kt_per_split = (a.ktiles + splits - 1) / splits;
  splits = (a.ktiles + a.kt_per_split - 1) / a.kt_per_split;
  dim3 grid((unsigned)(co_tiles * a.ci_tiles), (unsigned)taps, (unsigned)splits);
  hipLaunchKernelGGL((conv_wgrad_kernel<BCI, BCO, WCI_WAVES>), grid, dim3(256), 2 * 64 * (BCI + BCO) * 2, s, a);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

}  // namespace

extern "C" const char* danhip_conv_wgrad_kernel_label(const danhip_conv_desc* d) {
  if (!d) return "";
  const char* rl = danhip_wgrad_rows_label(d);
  if (rl) return rl;
  const char* pl = danhip_wgrad_pw_label(d);
  if (pl) return pl;
  const int co8 = (d->Cout + 7) / 8 * 8;
  if (danhip_wgrad_c8_eligible(d, 3, d->Cin, co8)) return "conv_wgrad_c8_kernel";
  if (d->Cin == 8 && d->kh * d->kw <= 16) return co8 > 64 ? "conv_wgrad_kernel<128, 128, 2>" : "conv_wgrad_kernel<128, 64, 2>";
  const bool ci_small = d->Cin <= 64, co_small = co8 <= 64;
  if (ci_small && co_small) return "conv_wgrad_kernel<64, 64, 2>";
  if (ci_small) return "conv_wgrad_kernel<64, 128, 2>";
  if (co_small) return "conv_wgrad_kernel<128, 64, 2>";
  return "conv_wgrad_kernel<128, 128, 2>";
}

extern "C" size_t danhip_conv2d_bwd_weight_workspace_bytes(const danhip_conv_desc* d) {
  if (!d) return 0;
  const int mode = danhip_option("wgrad_slab");
  if (!mode) return 0;
  const size_t r = danhip_wgrad_rows_workspace_bytes(d);
  return r ? r : danhip_wgrad_pw_workspace_bytes(d);
}

static int bwd_weight_impl(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio, float* db, int32_t cin_real, int ldx, int ldy,
                           void* ws, size_t ws_bytes, void* stream) {
  DH_REQUIRE(d && x && dy && dw_hwio, DANHIP_EINVAL, "conv2d_bwd_weight: null pointer");
  DH_REQUIRE(d->Cin % 8 == 0, DANHIP_EINVAL, "conv2d_bwd_weight: Cin=%d must be a multiple of 8", d->Cin);
  DH_REQUIRE(cin_real > 0 && cin_real <= d->Cin, DANHIP_EINVAL, "conv2d_bwd_weight: cin_real out of range");
  const int co8 = (d->Cout + 7) / 8 * 8;
  const bool view = ldx != d->Cin || ldy != co8;
  DH_REQUIRE(ldx >= d->Cin && ldy >= co8 && ((ldx | ldy) & 7) == 0, DANHIP_EINVAL, "conv2d_bwd_weight: pitches must be multiples of 8 and cover the channels");
  // the kernels address both activations through raw buffer descriptors with a 32-bit byte count: 2^31 16-bit elements = 4 GiB would wrap to 0
  DH_REQUIRE((int64_t)d->N * d->H * d->W * ldx < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo * (int64_t)ldy < (1ll << 31),
             DANHIP_EINVAL, "conv2d_bwd_weight: tensor exceeds 2^31 elements (4 GiB): split the batch");
  {   // output size: TF 'same' (ceil(in / s); padding derived, more on the bottom / right) or 'valid' (floor((in - k) / s) + 1, no padding)
    const bool same = d->Ho == (d->H + d->stride - 1) / d->stride && d->Wo == (d->W + d->stride - 1) / d->stride;
    const bool valid = d->H >= d->kh && d->W >= d->kw && d->Ho == (d->H - d->kh) / d->stride + 1 && d->Wo == (d->W - d->kw) / d->stride + 1;
    DH_REQUIRE(same || valid, DANHIP_EINVAL, "conv: Ho/Wo (%d,%d) is neither the 'same' nor the 'valid' output size", d->Ho, d->Wo);
  }
  {
    {
      const int hr = danhip_launch_wgrad_rows(d, x, dy, dw_hwio, db, cin_real, (hipStream_t)stream, ws, ws_bytes, ldx, ldy);
      if (hr <= 0) return hr;
    }
    const int pr = danhip_launch_wgrad_pw(d, x, dy, dw_hwio, db, cin_real, (hipStream_t)stream, ws, ws_bytes, ldx, ldy);
    if (pr <= 0) return pr;
  }
  WgradArgs a{};
  a.x = x; a.dy = dy; a.dw = dw_hwio; a.db = db;
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.Co8 = co8;
  a.ldx = ldx; a.ldy = ldy;
  a.cin_real = cin_real;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
  int total = (d->Ho - 1) * d->stride + d->kh - d->H; if (total < 0) total = 0; a.pad_t = total / 2;
  total = (d->Wo - 1) * d->stride + d->kw - d->W; if (total < 0) total = 0; a.pad_l = total / 2;
  a.M = d->N * d->Ho * d->Wo;
  a.ktiles = (a.M + 63) / 64;
  a.div_wo = make_fastdiv(a.Wo); a.div_howo = make_fastdiv(a.Ho * a.Wo);
  hipStream_t s = (hipStream_t)stream;
  if (danhip_wgrad_c8_eligible(d, cin_real, ldx, ldy))     // conv1_1: both operands staged once, taps as address offsets (conv_wgrad_c8.hip)
    return danhip_launch_wgrad_c8(d, x, dy, dw_hwio, db, cin_real, s);
  if (d->Cin == 8 && d->kh * d->kw <= 16 && !view) {       // first layer: taps ride in the tile columns
    a.tapcols = 1;
    return a.Co8 > 64 ? launch_wgrad<128, 128, 2>(a, s) : launch_wgrad<128, 64, 2>(a, s);
  }
  a.tapcols = 0;
  a.linear = d->kh == 1 && d->kw == 1 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W;
  const bool ci_small = d->Cin <= 64, co_small = a.Co8 <= 64;
  if (ci_small && co_small) return launch_wgrad<64, 64, 2>(a, s);
  if (ci_small) return launch_wgrad<64, 128, 2>(a, s);
  if (co_small) return launch_wgrad<128, 64, 2>(a, s);
  return launch_wgrad<128, 128, 2>(a, s);
}

extern "C" int danhip_conv2d_bwd_weight(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio, float* db,
                                        int32_t cin_real, void* stream) {
  return danhip_conv2d_bwd_weight_ws(d, x, dy, dw_hwio, db, cin_real, nullptr, 0, stream);
}

extern "C" int danhip_conv2d_bwd_weight_ws(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio, float* db,
                                           int32_t cin_real, void* ws, size_t ws_bytes, void* stream) {
  DH_REQUIRE(d != nullptr, DANHIP_EINVAL, "conv2d_bwd_weight: null pointer");
  return bwd_weight_impl(d, x, dy, dw_hwio, db, cin_real, d->Cin, (d->Cout + 7) / 8 * 8, ws, ws_bytes, stream);
}

/* x / dy as channel-slice views: pitch->x_pitch / y_pitch = elements between consecutive pixels of x / dy (aux_pitch unused). */
extern "C" int danhip_conv2d_bwd_weight_strided(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio, float* db,
                                                int32_t cin_real, const danhip_conv_pitch* pitch, void* ws, size_t ws_bytes, void* stream) {
  DH_REQUIRE(d != nullptr && pitch != nullptr, DANHIP_EINVAL, "conv2d_bwd_weight_strided: null pointer");
  return bwd_weight_impl(d, x, dy, dw_hwio, db, cin_real, pitch->x_pitch, pitch->y_pitch, ws, ws_bytes, stream);
}
