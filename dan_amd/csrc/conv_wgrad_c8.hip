// Weight gradient of the FIRST layer (conv1_1 of every backbone, net/sfd_net.py:128: 3x3 / stride 1 / 'same', the image padded to 8 channels
// of which cin_real <= 4 are real, 64 output channels):  dW[tap][c][co] += sum_px X[px @ tap][c] * dY[px][co],  db[co] += sum_px dY[px][co].
//
// The work is 0.7 GMAC per image against 52 MB of dY and 6.5 MB of X: HBM-bound by a wide margin (batch 16 at 640 x 640: 944 MB).  The
// general kernel (conv_wgrad.hip, "tapcols" mode) builds an im2col tile [64 px][9 taps x 8 ch] in LDS by 16-byte gathers - every pixel of X
// fetched nine times in 16-byte pieces - and ran at 3.1 TB/s (306 us at batch 16, ALONE at the very end of the backward pass: nothing is
// left to run beside it).  Here both operands are staged exactly as they lie in HBM, each ONCE:
//   * a persistent 512-thread workgroup per CU walks tiles of 8 rows x 64 pixels; per tile the (8+2) x (64+2)-pixel halo patch of X (16 bytes
//     per pixel: 10.3 KiB) and the 8 x 64 x 64-channel tile of dY (64 KiB) are DMA'd into one of two LDS buffers while the other is consumed;
//   * wave w owns row w of the tile and accumulates the WHOLE gradient (M = 12 tap slots x 4 channels = three 16-row MFMA tiles, N = 64)
//     over its pixels in 48 accumulator registers; the reduction axis of both operands is the PIXEL (the slow axis of NHWC), so both fragments
//     come from the gfx950 transposing read ds_read_b64_tr_b16, whose per-lane row addresses make the nine taps plain address offsets into
//     the one patch (tap (i, j) of pixel p = patch pixel p + i * 66 + j) - no im2col anywhere;
//   * tap slot 9 reads a constant [1, 0, 0, 0]: its channel-0 row is the bias gradient; slots 10, 11 read zeros;
//   * at the end the eight waves' gradients are summed in LDS and leave the CU as 1728 + 64 fp32 atomics.
#include "conv_common.h"

namespace {

struct WgC8Args {
  const bf16_t* x;     // [N,H,W,8]
  const bf16_t* dy;    // [N,H,W,64]
  float* dw;           // [3,3,cin_real,64]
  float* db;           // [64] or null
  int N, H, W, cin_real;
  int tiles_x, tiles_y, items;
  FastDiv div_tx, div_txy;
};

constexpr int C8_TH = 8, C8_TW = 64, C8_PW = C8_TW + 2;
constexpr int C8_PROWS = (C8_TH + 2) * C8_PW;                 // 660 patch pixels
constexpr int C8_XPIECES = (C8_PROWS + 63) / 64;              // 11 DMA pieces of 64 pixels x 16 bytes
constexpr int C8_XBYTES = C8_XPIECES * 1024;
constexpr int C8_YBYTES = C8_TH * C8_TW * 128;                // 64 KiB
constexpr int C8_STAGE = C8_XBYTES + C8_YBYTES;
constexpr int C8_CONST = 2 * C8_STAGE;                        // 16 bytes: [1, 0, 0, 0 | 0, 0, 0, 0]
constexpr int C8_LDS = C8_CONST + 64;

__global__ __launch_bounds__(512, 2) void conv_wgrad_c8_kernel(const WgC8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int G = gridDim.x;

  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)((unsigned)(a.N * a.H * a.W) * 16u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dy), 0, (int)((unsigned)(a.N * a.H * a.W) * 128u), 0x00020000);
  auto coords = [&](int it, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    n = (int)fdiv((unsigned)it, a.div_txy);
    const int rem = it - n * (a.tiles_x * a.tiles_y);
    const int ty = (int)fdiv((unsigned)rem, a.div_tx);
    y0 = ty * C8_TH;
    x0 = (rem - ty * a.tiles_x) * C8_TW;
  };
  auto issue = [&](int it, int buf) __attribute__((always_inline)) {
    int n, y0, x0;
    coords(it, n, y0, x0);
    char* sx = smem + buf * C8_STAGE;
    char* sy = sx + C8_XBYTES;
#pragma unroll
    for (int k = 0; k < 2; ++k) {                              // X patch: 11 pieces over 8 waves (the last one is loaded twice: uniform DMA count)
      int piece = k * 8 + wave;
      if (piece > C8_XPIECES - 1) piece = C8_XPIECES - 1;
      const int e = piece * 64 + lane;
      const int r = e / C8_PW, c = e - r * C8_PW;
      const int y = y0 - 1 + r, x = x0 - 1 + c;
      const bool ok = e < C8_PROWS && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_AS void*)(sx + piece * 1024), 16, ok ? (unsigned)((n * a.H + y) * a.W + x) * 16u : 0xFFFFFFFFu, 0, 0,
                                               0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {                              // dY tile: 64 pieces of 8 pixels x 128 bytes; wave w loads its own row
      const int tp = (wave * 8 + k) * 8 + (lane >> 3);         // pixel of the tile: row tp / 64 (= wave), column tp % 64
      const int y = y0 + (tp >> 6), x = x0 + (tp & 63);
      const bool ok = y < a.H && x < a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (LDS_AS void*)(sy + (wave * 8 + k) * 1024), 16,
                                               ok ? (unsigned)((n * a.H + y) * a.W + x) * 128u + (unsigned)((lane & 7) << 4) : 0xFFFFFFFFu, 0, 0, 0);
    }
  };

  if (tid < 16) reinterpret_cast<bf16_t*>(smem + C8_CONST)[tid] = tid == 0 ? f2bf(1.0f) : (bf16_t)0;
  int it = blockIdx.x;
  if (it >= a.items) return;                                   // (uniform)
  issue(it, 0);

  // per-lane fragment addresses inside a stage (+ ks * 32 pixels, + h * 4 pixels): lane 4q+p of a 16-lane group supplies row q (= pixel) of
  // the transposed 4 x 16 block, columns 4p .. 4p+3 (= tap slot p of the M tile, channels 0..3 | output channels 4p .. 4p+3 of the N tile)
  int xaddr[3];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    const int tap = mt * 4 + p;
    const int ti = tap / 3, tj = tap - ti * 3;
    xaddr[mt] = tap < 9 ? ((wave + ti) * C8_PW + 8 * g + q + tj) * 16 : (tap == 9 ? -1 : -2);
  }
  const int yaddr = C8_XBYTES + ((wave * 64 + 8 * g + q) * 128) + p * 8;

  f32x4 acc[3][4];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (;;) {
    const int nxt = it + G;
    const bool has_next = nxt < a.items;
    if (has_next) issue(nxt, buf ^ 1);                         // (the other buffer was released by the barrier that ended the previous tile)
    const char* st = smem + buf * C8_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x4 xh[2][3], yh[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int po = (ks * 32 + 4 * h);                      // pixel offset of this half inside the row
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          const char* ad = xaddr[mt] >= 0 ? st + xaddr[mt] + po * 16 : smem + C8_CONST + (xaddr[mt] == -1 ? 0 : 8);
          xh[h][mt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)ad);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) yh[h][nt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(st + yaddr + po * 128 + nt * 32));
      }
      // (whole-vector concatenation + bitcast: element-wise extraction of the tr-read result miscompiles on ROCm 7.2, conv_wgrad.hip)
      bf16x8 xf[3], yf[4];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) xf[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xh[0][mt], xh[1][mt], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) yf[nt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(yh[0][nt], yh[1][nt], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = DH_MFMA_16x16x32(xf[mt], yf[nt], acc[mt][nt]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next tile has landed (own pieces) ...
    __syncthreads();                                           // ... and everybody is done reading this one
    if (!has_next) break;
    it = nxt;
    buf ^= 1;
  }

  // ---- eight waves -> one gradient in LDS -> global atomics.  Lane holds acc[mt][nt][r] = dW[tap slot mt*4 + g][channel r][co = nt*16 + (lane & 15)]
  float* red = reinterpret_cast<float*>(smem);                 // [12 slots][4 channels][64 co]
  for (int i = tid; i < 12 * 4 * 64; i += 512) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[((mt * 4 + g) * 4 + r) * 64 + nt * 16 + (lane & 15)], acc[mt][nt][r]);
  __syncthreads();
  for (int i = tid; i < 9 * 4 * 64; i += 512) {
    const int co = i & 63, c = (i >> 6) & 3, slot = i >> 8;
    if (c < a.cin_real) atomicAdd(a.dw + (size_t)(slot * a.cin_real + c) * 64 + co, red[i]);
  }
  if (a.db && tid < 64) atomicAdd(a.db + tid, red[(9 * 4 + 0) * 64 + tid]);
}

}  // namespace

bool danhip_wgrad_c8_eligible(const danhip_conv_desc* d, int cin_real, int ldx, int ldy) {
  return d->Cin == 8 && cin_real >= 1 && cin_real <= 4 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W && d->Cout == 64 &&
         ldx == 8 && ldy == 64 && (int64_t)d->N * d->H * d->W * 128 < (1ll << 32) && danhip_option("wgrad_c8") != 0;
}

int danhip_launch_wgrad_c8(const danhip_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* db, int cin_real, hipStream_t s) {
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_c8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS) == hipSuccess;
  (void)attr_ok;
  WgC8Args a{};
  a.x = x; a.dy = dy; a.dw = dw; a.db = db;
  a.N = d->N; a.H = d->H; a.W = d->W; a.cin_real = cin_real;
  a.tiles_x = (d->W + C8_TW - 1) / C8_TW;
  a.tiles_y = (d->H + C8_TH - 1) / C8_TH;
  a.items = d->N * a.tiles_x * a.tiles_y;
  a.div_tx = make_fastdiv(a.tiles_x);
  a.div_txy = make_fastdiv(a.tiles_x * a.tiles_y);
  int G = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) G = v;
  }
  if (a.items < G) G = a.items;
  hipLaunchKernelGGL(conv_wgrad_c8_kernel, dim3(G), dim3(512), C8_LDS, s, a);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}
