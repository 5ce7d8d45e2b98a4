// Implicit-GEMM convolution on MFMA for gfx950 (bf16 in, fp32 accumulate), NHWC.
//
// Replaces tf.layers.conv2d(padding='same') on the reference's hot path (net/sfd_net.py:81-89 and every conv in
// net/*.py) and, run on dY with tap-flipped weights, its data gradient.
//
// GEMM view:  out[m, co] = sum_k  A[m, k] * Wp[co, k],   m = (n, ho, wo),  k = (tap, c)
//   A[m,k] is gathered on the fly from the NHWC input (im2col never materialised): one 16-byte LDS-DMA
//   (global_load_lds_dwordx4) per lane moves 8 consecutive channels of one (pixel, tap); padding and
//   out-of-range rows read a page of zeros.  Both operands are K-contiguous, so both tiles are
//   [rows][64 bf16] = 128-byte rows in LDS, XOR-swizzled at 16-byte granularity (chunk ^= row & 7; applied on
//   the SOURCE address because the LDS-DMA destination is lane-linear) so every ds_read_b128 is conflict-free.
//   MFMA: v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand and the PIXEL fragment as B, so
//   each lane ends up with 4 consecutive output channels of one pixel (8/16-byte stores into NHWC).
//
// Pipeline: two LDS stages; tile k+1's DMA is issued before tile k's MFMAs (one barrier per K tile);
// two workgroups per CU overlap each other's barrier stalls.
#include <cstdlib>

#include "conv_common.h"

namespace {


template <int BM, int BN, int WN_WAVES, bool FAST>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int RA = BM / 32;                 // LDS-DMA rounds for the pixel tile (each round: 4 waves x 8 rows)
  constexpr int RB = (BN + 31) / 32;          // rounds for the weight tile
  constexpr int STAGE = (BM + BN) * 128;      // bytes per pipeline stage
  constexpr int WM_WAVES = 4 / WN_WAVES;
  constexpr int TP = BM / WM_WAVES;           // pixels per wave
  constexpr int TC = BN / WN_WAVES;           // output channels per wave
  constexpr int NPT = TP / 16, NCT = TC / 16;
  static_assert(TP % 16 == 0 && TC % 16 == 0, "wave tile must be MFMA-tile aligned");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;

  // ---- tile coordinates: blockIdx.x walks pixel tiles (XCD-friendly: neighbouring pixel tiles share halo rows),
  //      blockIdx.y walks output-channel tiles.
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-lane staging geometry (constant over the K loop)
  const int srow = lane >> 3;                          // row inside an 8-row DMA piece
  const int schunk = (lane & 7) ^ srow;                // source 16-byte chunk (XOR swizzle on the source side)
  int rn[RA], rh[RA], rw[RA];                          // per staged pixel row: image base offset, h0, w0
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + j * 32 + wave * 8 + srow;
    if (m < a.M) {
      const unsigned n = fdiv((unsigned)m, a.div_howo);
      const unsigned rem = (unsigned)m - n * (unsigned)(a.Ho * a.Wo);
      const unsigned ho = fdiv(rem, a.div_wo);
      const unsigned wo = rem - ho * (unsigned)a.Wo;
      rn[j] = (int)n * a.H * a.W;                      // pixel index of (n,0,0)
      rh[j] = (int)ho * a.stride - a.pad_t;
      rw[j] = (int)wo * a.stride - a.pad_l;
    } else {
      rn[j] = 0; rh[j] = -(1 << 20); rw[j] = 0;        // never valid
    }
  }
  const bf16_t* wrow[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    int r = j * 32 + wave * 8 + srow;
    if (BN < 32 && r >= BN) r = 0;                     // (those waves skip the DMA below)
    wrow[j] = a.w + (size_t)(n0 + r) * a.Kpad + schunk * 8;
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_danhip_zero_page);

  // split-K: blockIdx.z owns the K tiles [kt_lo, kt_hi) and stores raw fp32 partial sums (conv_splitk_finish_kernel applies the epilogue)
  const int kt_lo = a.splitk_ws ? (int)blockIdx.z * a.kt_per_split : 0;
  const int kt_hi = a.splitk_ws ? min(a.ktiles, kt_lo + a.kt_per_split) : a.ktiles;

  // FAST path tap walker (uniform): tile kt covers channels [c0, c0+64) of tap (ti, tj)
  int ti = 0, tj = 0, c0 = 0;

  // FAST + stride-1 gather (every forward conv and every stride-1 data gradient): per staged row one 32-bit byte offset of the
  // tap-(0,0) source pixel and a bit mask of the taps that fall inside the image; a tile then adds a wave-uniform tap/channel
  // offset (computed on the scalar ALU) and selects 0xFFFFFFFF for padding taps (the buffer descriptor's range check zero-fills)
  // — 4 VALU per row and tile instead of ~15 (64-bit address build + zero-page select).
  const bool lean = FAST && a.dstride == 1 && a.taps <= 32;
  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)(((unsigned)(a.N * a.H * a.W - 1) * (unsigned)a.ldx + (unsigned)a.C) * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)((unsigned)(gridDim.y * BN) * (unsigned)a.Kpad * 2u), 0x00020000);
  unsigned pbase[RA];
  unsigned tmask[RA];
  unsigned wvoff[RB];
  if (lean) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      pbase[j] = (unsigned)((rn[j] + rh[j] * a.W + rw[j]) * a.ldx + schunk * 8) * 2u;        // may wrap for padding rows; exact for valid taps
      unsigned mk = 0;
      for (int t = 0; t < a.taps; ++t) {
        const int i = (int)fdiv((unsigned)t, a.div_kw), jj = t - i * a.kw;
        if ((unsigned)(rh[j] + i) < (unsigned)a.H && (unsigned)(rw[j] + jj) < (unsigned)a.W) mk |= 1u << t;
      }
      tmask[j] = mk;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      int r = j * 32 + wave * 8 + srow;
      if (BN < 32 && r >= BN) r = 0;
      wvoff[j] = (unsigned)((n0 + r) * a.Kpad + schunk * 8) * 2u;
    }
  }
  int tap = 0;
  if (FAST && kt_lo > 0) {                               // (C % 64 == 0: a K tile never straddles two taps)
    tap = (int)fdiv((unsigned)(kt_lo * 64), a.div_c);
    c0 = kt_lo * 64 - tap * a.C;
    ti = (int)fdiv((unsigned)tap, a.div_kw);
    tj = tap - ti * a.kw;
  }

  auto stage = [&](int kt, int buf) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + BM * 128;
    if (FAST && lean) {
      const unsigned toff = (unsigned)((ti * a.W + tj) * a.ldx + c0) * 2u;                // wave-uniform
      const unsigned tbit = 1u << tap;
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const unsigned voff = (tmask[j] & tbit) ? pbase[j] + toff : 0xFFFFFFFFu;
        bufdma16_lds(rsrc_x, voff, 0u, sA + (j * 32 + wave * 8) * 128);
      }
    } else if (FAST) {
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        int hi = rh[j] + ti, wi = rw[j] + tj;
        bool ok = true;
        if (a.dstride > 1) {                            // transposed (fractionally strided) gather
          ok = ((hi | wi) & (a.dstride - 1)) == 0 && hi >= 0 && wi >= 0;
          hi >>= a.dshift; wi >>= a.dshift;
        }
        ok = ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const bf16_t* src = ok ? a.x + ((size_t)(rn[j] + hi * a.W + wi) * a.ldx + c0 + schunk * 8) : zero;
        glds16(src, sA + (j * 32 + wave * 8) * 128);
      }
    } else {
      const unsigned e = (unsigned)(kt * 8 + schunk) * 8u;   // first k element of this lane's chunk
      const unsigned tap = fdiv(e, a.div_c);
      const int cc = (int)(e - tap * (unsigned)a.C);
      const int i = (int)fdiv(tap, a.div_kw);
      const int jj = (int)tap - i * a.kw;
      const bool tapok = (int)tap < a.taps;
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        int hi = rh[j] + i, wi = rw[j] + jj;
        bool ok = tapok;
        if (a.dstride > 1) {
          ok = ok && ((hi | wi) & (a.dstride - 1)) == 0 && hi >= 0 && wi >= 0;
          hi >>= a.dshift; wi >>= a.dshift;
        }
        ok = ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const bf16_t* src = ok ? a.x + ((size_t)(rn[j] + hi * a.W + wi) * a.ldx + cc) : zero;
        glds16(src, sA + (j * 32 + wave * 8) * 128);
      }
    }
    if (FAST && lean) {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        if (BN >= 32 || (j * 32 + wave * 8) < BN)
          bufdma16_lds(rsrc_w, wvoff[j], (unsigned)kt * 128u, sB + (j * 32 + wave * 8) * 128);
      }
    } else {
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        if (BN >= 32 || (j * 32 + wave * 8) < BN) glds16(wrow[j] + kt * 64, sB + (j * 32 + wave * 8) * 128);
      }
    }
    if (FAST) {                                         // advance the tap walker
      c0 += 64;
      if (c0 == a.C) { c0 = 0; ++tap; if (++tj == a.kw) { tj = 0; ++ti; } }
    }
  };

  f32x4 acc[NCT][NPT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int p = 0; p < NPT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;

  stage(kt_lo, kt_lo & 1);
  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // tile kt landed; everyone is done reading the other stage
    if (kt + 1 < kt_hi) stage(kt + 1, (kt + 1) & 1);
    const char* sA = smem + (kt & 1) * STAGE;
    const char* sB = sA + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[NCT], xf[NPT];
      const int q = ks * 4 + fq;
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int r = wn * TC + c * 16 + frow;
        wf[c] = *reinterpret_cast<const bf16x8*>(sB + r * 128 + ((q ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const int r = wm * TP + p * 16 + frow;
        xf[p] = *reinterpret_cast<const bf16x8*>(sA + r * 128 + ((q ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int p = 0; p < NPT; ++p) acc[c][p] = DH_MFMA_16x16x32(wf[c], xf[p], acc[c][p]);
    }
  }

  // ---- epilogue: lane holds out[pixel = p*16 + (lane&15)][co = c*16 + (lane>>4)*4 + 0..3]
  if (a.splitk_ws) {
    float* slab = a.splitk_ws + (size_t)blockIdx.z * (size_t)a.M * a.Co;
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const int m = m0 + wm * TP + p * 16 + frow;
      if (m >= a.M) continue;
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int co = n0 + wn * TC + c * 16 + fq * 4;
        if (co >= a.Co) continue;
        float* o = slab + (size_t)m * a.Co + co;
        if ((co + 4 <= a.Co) && ((a.Co & 3) == 0)) {
          *reinterpret_cast<f32x4*>(o) = acc[c][p];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (co + r < a.Co) o[r] = acc[c][p][r];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int m = m0 + wm * TP + p * 16 + frow;
    if (m >= a.M) continue;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int co = n0 + wn * TC + c * 16 + fq * 4;
      if (co >= a.Co) continue;
      float v[4] = {acc[c][p][0], acc[c][p][1], acc[c][p][2], acc[c][p][3]};
      const size_t o = (size_t)m * (a.split_out ? 3 * a.ldy : a.ldy) + co;          // (pitched views: conv_common.h ConvArgs::ldx / ldy / ldm)
      const size_t om = (size_t)m * a.ldm + co, orr = (size_t)m * a.Co + co;
      const bool full = (co + 4 <= a.Co) && ((a.Co & 3) == 0);
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (co + r < a.Co) v[r] += a.bias[co + r];
      }
      if (a.relu && co < a.relu_co) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dh_relu(v[r]);
      }
      if (a.out_f32) {
        float* y = reinterpret_cast<float*>(a.y) + o;
        if (full) {
          float4 t = make_float4(v[0], v[1], v[2], v[3]);
          if (a.accumulate) { float4 u = *reinterpret_cast<float4*>(y); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
          *reinterpret_cast<float4*>(y) = t;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (co + r < a.Co) y[r] = a.accumulate ? y[r] + v[r] : v[r];
        }
      } else {
        bf16_t* y = reinterpret_cast<bf16_t*>(a.y) + o;
        if (a.split_out) {                                // the next convolution's [hi | lo | hi] limb layout (split_infer.hip)
          uint2 hi, lo;
          dh_split4(v, hi, lo);
          *reinterpret_cast<uint2*>(y) = hi;
          *reinterpret_cast<uint2*>(y + a.Co) = lo;
          *reinterpret_cast<uint2*>(y + 2 * a.Co) = hi;
          continue;
        }
        if (full) {
          if (a.mask) {
            const uint2 mk = *reinterpret_cast<const uint2*>(a.mask + om);
            const bf16_t* mp = reinterpret_cast<const bf16_t*>(&mk);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (!(bf2f(mp[r]) > 0.f)) v[r] = 0.f;
          }
          if (a.resid) {
            const uint2 rs = *reinterpret_cast<const uint2*>(a.resid + orr);
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
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (co + r < a.Co) {
              float t = v[r];
              if (a.mask && !(bf2f(a.mask[om + r]) > 0.f)) t = 0.f;
              if (a.resid) t += bf2f(a.resid[orr + r]);
              if (a.accumulate) t += bf2f(y[r]);
              y[r] = f2bf(t);
            }
          }
        }
      }
    }
  }
}

// Second pass of a split-K launch: one thread per 4 output channels of a pixel sums the partial outputs and applies the epilogue
// (bias, ReLU, mask, residual, accumulate, output type) exactly as the single-pass kernels do (conv_store4).
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvArgs a) {
  const int cq = (a.Co + 3) >> 2;
  const long total = (long)a.M * cq;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const long m = gid / cq;
  const int co = (int)(gid - m * cq) * 4;
  const size_t slab = (size_t)a.M * a.Co;
  const float* src = a.splitk_ws + (size_t)m * a.Co + co;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if ((a.Co & 3) == 0) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int z = 0;
    for (; z + 2 <= a.splits; z += 2) {
      s0 += *reinterpret_cast<const f32x4*>(src + (size_t)z * slab);
      s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(z + 1) * slab);
    }
    if (z < a.splits) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)z * slab);
    s0 += s1;
    v[0] = s0[0]; v[1] = s0[1]; v[2] = s0[2]; v[3] = s0[3];
  } else {
    for (int z = 0; z < a.splits; ++z)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (co + r < a.Co) v[r] += src[(size_t)z * slab + r];
  }
  conv_store4(a, v, (size_t)m, co);
}

// Tile selection: output-channel tile BN from the (padded) channel count.
inline int pick_bn(int co) {
  if (co % 128 == 0) return 128;
  if (co % 64 == 0) return 64;
  if (co <= 16) return 16;
  if (co <= 32) return 32;
  return 64;
}

int igemm_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

// Pixel tile of the flat-M configuration launch_conv picks for these args.
inline int igemm_bm(const ConvArgs& a) {
  const int bn = pick_bn(a.Co);
  if (bn == 128) return 128;
  if (bn == 16) return a.M <= 256 * 128 ? 64 : 256;
  return 256;
}

// Split-K plan: maps with too few output tiles to fill the chip (the 40x40 ... 5x5 levels, every level at 2-4 images per GPU) walk K
// (up to 9 taps x 1024 channels = 144 tiles) serially in a handful of workgroups; splitting K over blockIdx.z puts ~2 workgroups on every
// CU.  Returns the number of splits (1 = do not split) and the K tiles per split.
int plan_splitk(const ConvArgs& a, int* kt_per_split) {
  *kt_per_split = a.ktiles;
  if (!danhip_option("splitk") || a.pool_y || a.bits_out || a.mask_bits) return 1;        // (DANHIP_SPLITK=0 / danhip_set_option: A/B)
  const int bm = igemm_bm(a), bn = pick_bn(a.Co);
  const long tiles = (long)cdiv(a.M, bm) * cdiv(a.Co, bn);
  const int cus = igemm_cu_count();
  // thin heads (16-channel tiles): a workgroup's K tile is one or two MFMAs per wave, eight workgroups fit a CU and the serial K walk
  // (72-144 tiles at ~0.6 us) is pure latency: aim for 8 workgroups per CU instead of 2
  const long target = (bn == 16 ? 8l : 2l) * cus;
  if (tiles * 2 > target || a.ktiles < 8) return 1;
  int splits = (int)((target + tiles - 1) / tiles);
  if (splits > a.ktiles / 4) splits = a.ktiles / 4;
  if (splits > 36) splits = 36;
  if (splits < 2) return 1;
  const int per = (a.ktiles + splits - 1) / splits;
  *kt_per_split = per;
  return (a.ktiles + per - 1) / per;
}

// true when a split-K flat-M launch should take precedence over the streaming kernels (their 128-pixel items would not fill the chip)
bool wants_splitk(const ConvArgs& a) {
  int per;
  if (plan_splitk(a, &per) < 2) return false;
  return (long)cdiv(a.M, 128) * cdiv(a.Co, 128) * 2 <= igemm_cu_count();
}
bool prefer_splitk(const ConvArgs& a) { return a.splitk_ws && wants_splitk(a); }

template <int BM, int BN, int WN_WAVES>
int launch_cfg(const ConvArgs& a0, bool fast, hipStream_t s) {
  ConvArgs a = a0;
  int per = a.ktiles;
  const int splits = a.splitk_ws ? plan_splitk(a, &per) : 1;
  if (splits < 2) a.splitk_ws = nullptr;
  a.splits = splits; a.kt_per_split = per;
  dim3 grid((unsigned)cdiv(a.M, BM), (unsigned)cdiv(a.Co, BN), (unsigned)splits);
  const size_t lds = 2 * (size_t)(BM + BN) * 128;
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WN_WAVES, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(2 * (BM + BN) * 128)) == hipSuccess &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WN_WAVES, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(2 * (BM + BN) * 128)) == hipSuccess;
  (void)attr_ok;
  if (fast)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WN_WAVES, true>), grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WN_WAVES, false>), grid, dim3(256), lds, s, a);
  DH_LAUNCH_CHECK();
  if (a.splitk_ws) {
    const long threads = (long)a.M * ((a.Co + 3) / 4);
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    DH_LAUNCH_CHECK();
  }
  return DANHIP_OK;
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
  const bool view = a.strided();                       // channel-slice views / partial ReLU: the streaming GEMM and the flat-M kernel only
  const bool limbs = a.split_out != 0;                 // limb-layout output: the halo kernel's general epilogue or the flat-M kernel (conv_store4)
  if (!view && !limbs) {
    const int c8 = danhip_launch_conv_c8(a, s);        // conv1_1: 3 (padded to 8) -> 64 channels, bound by its output write
    if (c8 <= 0) return c8;
  }
  if (!limbs) {
    const int cr = danhip_launch_conv_c64(a, s);       // 3x3 / stride-1, 64 -> 64 channels: register-resident weights (views too)
    if (cr <= 0) return cr;
  }
  const bool sk = prefer_splitk(a);                    // too few tiles for the persistent kernels: split K over workgroups instead
  if (!sk) {
    if (!view) {
      const int hr = danhip_launch_conv_halo(a, s);    // 3x3 / stride-1 on large maps: halo-reuse kernel
      if (hr <= 0) return hr;
    }
    if (!limbs) {
      const int pr = danhip_launch_conv_pointwise(a, s); // 1x1 / stride-1 with 64-multiple channels: streaming GEMM
      if (pr <= 0) return pr;
    }
  }
  const bool fast = (a.C % 64 == 0);
  switch (pick_bn(a.Co)) {
    case 128: return launch_cfg<128, 128, 2>(a, fast, s);
    case 64: return launch_cfg<256, 64, 1>(a, fast, s);
    case 32: return launch_cfg<256, 32, 1>(a, fast, s);
    default:
      // thin heads on small maps: 256-pixel tiles would leave most CUs idle (20x20x16 images = 25 tiles) -> 64-pixel tiles
      if (a.M <= 256 * 128) return launch_cfg<64, 16, 1>(a, fast, s);
      return launch_cfg<256, 16, 1>(a, fast, s);
  }
}

inline int same_pad_before(int in, int out, int k, int s) {
  int total = (out - 1) * s + k - in;
  if (total < 0) total = 0;
  return total / 2;
}

int check_desc(const danhip_conv_desc* d) {
  DH_REQUIRE(d != nullptr, DANHIP_EINVAL, "conv: null descriptor");
  DH_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, DANHIP_EINVAL, "conv: non-positive dims");
  DH_REQUIRE(d->Cin % 8 == 0, DANHIP_EINVAL, "conv: Cin=%d must be a multiple of 8 (pad the activation)", d->Cin);
  DH_REQUIRE(d->kh >= 1 && d->kh <= 7 && d->kw >= 1 && d->kw <= 7, DANHIP_EINVAL, "conv: kernel %dx%d unsupported", d->kh, d->kw);
  DH_REQUIRE(d->stride >= 1 && d->stride <= 4, DANHIP_EINVAL, "conv: stride %d unsupported", d->stride);
  {   // output size: TF 'same' (ceil(in / s); padding derived, more on the bottom / right) or 'valid' (floor((in - k) / s) + 1, no padding)
    const bool same = d->Ho == (d->H + d->stride - 1) / d->stride && d->Wo == (d->W + d->stride - 1) / d->stride;
    const bool valid = d->H >= d->kh && d->W >= d->kw && d->Ho == (d->H - d->kh) / d->stride + 1 && d->Wo == (d->W - d->kw) / d->stride + 1;
    DH_REQUIRE(same || valid, DANHIP_EINVAL, "conv: Ho/Wo (%d,%d) is neither the 'same' nor the 'valid' output size", d->Ho, d->Wo);
  }
  DH_REQUIRE((int64_t)d->N * d->H * d->W * d->Cin < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo * (int64_t)((d->Cout + 7) / 8 * 8) < (1ll << 31),
             DANHIP_EINVAL, "conv: tensor exceeds 2^31 elements");
  return DANHIP_OK;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" int danhip_conv_packed_dims(const danhip_conv_desc* d, int which, int64_t* rows, int64_t* cols) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(rows && cols, DANHIP_EINVAL, "conv_packed_dims: null output");
  const int taps = d->kh * d->kw;
  if (which == 0) {
    *rows = round_up(d->Cout, pick_bn(d->Cout));
    *cols = round_up(taps * d->Cin, 64);
  } else {
    const int co8 = round_up(d->Cout, 8);
    *rows = round_up(d->Cin, pick_bn(d->Cin));
    *cols = round_up(taps * co8, 64);
  }
  return DANHIP_OK;
}

namespace {
// ---- weight packing.  Source: HWIO fp32 w[tap][c][co] (co contiguous).  Forward packing wf[co][k = tap*cin + c] is a TRANSPOSE of it:
// an element-wise kernel reads the source at a stride of cout floats (one 64-byte sector per 4 bytes used — 1.2 ms per PyramidBox step);
// here a workgroup moves a 64 k x 32 co tile through LDS: source rows read as 128-byte lines, packed rows written as 16-byte vectors.
// Backward packing wb[ci][k = tap_flipped*co8 + co] keeps co contiguous: element-wise with 32-bit index arithmetic.
// An entry's workgroups: the first e.pad_ ( = forward tiles) do the tiles, the rest stride over the backward elements.
__device__ __forceinline__ void pack_entry_block(const danhip_pack_entry& e, int lb, int nb, float (*tile)[33]) {
  const float* __restrict__ w = e.w_hwio;
  const int taps = e.kh * e.kw;
  const int tid = threadIdx.x;
  if (lb < e.pad_) {
    bf16_t* __restrict__ wf = e.wf_packed;
    const int ctiles = (e.rows_f + 31) / 32;
    const int tco = lb % ctiles, tk = lb / ctiles;
    const int co0 = tco * 32, k0 = tk * 64;
    const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int kl = ty + 8 * r, k = k0 + kl;
      const int tap = k / e.cin, c = k - tap * e.cin;
      float v = 0.f;
      if (co0 + tx < e.cout && tap < taps && c < e.cin_real) v = w[((long)tap * e.cin_real + c) * e.cout + co0 + tx];
      tile[kl][tx] = v;
    }
    __syncthreads();
    const int col = tid >> 3, kq = tid & 7;           // 32 output rows x eight 16-byte pieces
    if (co0 + col < e.rows_f) {
      uint4 o;
      o.x = pack2bf(tile[kq * 8 + 0][col], tile[kq * 8 + 1][col]);
      o.y = pack2bf(tile[kq * 8 + 2][col], tile[kq * 8 + 3][col]);
      o.z = pack2bf(tile[kq * 8 + 4][col], tile[kq * 8 + 5][col]);
      o.w = pack2bf(tile[kq * 8 + 6][col], tile[kq * 8 + 7][col]);
      *reinterpret_cast<uint4*>(wf + (long)(co0 + col) * e.cols_f + k0 + kq * 8) = o;
    }
    return;
  }
  bf16_t* __restrict__ wb = e.wb_packed;
  if (!wb) return;
  const unsigned total_b = (unsigned)e.rows_b * (unsigned)e.cols_b;      // (< 2^31: danhip_pack_entry_init checks)
  const unsigned nbb = (unsigned)(nb - e.pad_);
  for (unsigned j = (unsigned)(lb - e.pad_) * 256u + tid; j < total_b; j += nbb * 256u) {
    const unsigned ci = j / (unsigned)e.cols_b, k = j - ci * (unsigned)e.cols_b;
    const unsigned tapf = k / (unsigned)e.co8, co = k - tapf * (unsigned)e.co8;          // flipped tap index
    float v = 0.f;
    if ((int)ci < e.cin_real && (int)tapf < taps && (int)co < e.cout) {
      const int fi = tapf / e.kw, fj = tapf - fi * e.kw;
      const int tap = (e.kh - 1 - fi) * e.kw + (e.kw - 1 - fj);
      v = w[((long)tap * e.cin_real + ci) * e.cout + co];
    }
    wb[j] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const danhip_pack_entry e) {
  __shared__ float tile[64][33];
  pack_entry_block(e, (int)blockIdx.x, (int)gridDim.x, tile);
}

// All conv weights of a model in ONE launch (the per-layer launches cost more than the packing itself: 28 per S3FD step,
// 230 per DAN step).  Block b works on entry e = the last one with first_block[e] <= b (binary search over <= a few hundred).
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const danhip_pack_entry* __restrict__ tab, int n) {
  __shared__ float tile[64][33];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const danhip_pack_entry e = tab[lo];
  const int nb = (lo + 1 < n ? tab[lo + 1].first_block : (int)gridDim.x) - e.first_block;
  pack_entry_block(e, (int)blockIdx.x - e.first_block, nb, tile);
}
}  // namespace



namespace {
// ConvArgs of a forward call / of a stride-1 (or power-of-two strided) data-gradient call of descriptor d.
ConvArgs fwd_args(const danhip_conv_desc* d) {
  ConvArgs a{};
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Cout;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
  a.pad_t = same_pad_before(d->H, d->Ho, d->kh, d->stride);
  a.pad_l = same_pad_before(d->W, d->Wo, d->kw, d->stride);
  a.M = d->N * d->Ho * d->Wo;
  a.taps = d->kh * d->kw;
  a.Kpad = round_up(a.taps * a.C, 64);
  a.ktiles = a.Kpad / 64;
  a.cpt = a.C / 64;
  a.div_wo = make_fastdiv(a.Wo); a.div_howo = make_fastdiv(a.Ho * a.Wo); a.div_c = make_fastdiv(a.C); a.div_kw = make_fastdiv(a.kw);
  a.dstride = 1; a.dshift = 0;
  a.ldx = a.C; a.ldy = a.Co; a.ldm = a.Co; a.relu_co = a.Co;
  return a;
}
ConvArgs bwd_args(const danhip_conv_desc* d) {
  ConvArgs a{};
  const int co8 = round_up(d->Cout, 8);
  const int taps = d->kh * d->kw;
  const int pad_t = same_pad_before(d->H, d->Ho, d->kh, d->stride), pad_l = same_pad_before(d->W, d->Wo, d->kw, d->stride);
  // a forward convolution of dy with the tap-flipped, transposed weights; pad' = k-1-pad
  a.N = d->N; a.H = d->Ho; a.W = d->Wo; a.C = co8; a.Ho = d->H; a.Wo = d->W; a.Co = d->Cin;
  a.kh = d->kh; a.kw = d->kw; a.stride = 1;
  a.pad_t = d->kh - 1 - pad_t; a.pad_l = d->kw - 1 - pad_l;
  a.dstride = d->stride; a.dshift = 0;
  while ((1 << a.dshift) < d->stride) ++a.dshift;
  a.M = d->N * d->H * d->W;
  a.taps = taps;
  a.Kpad = round_up(taps * co8, 64);
  a.ktiles = a.Kpad / 64;
  a.cpt = a.C / 64;
  a.div_wo = make_fastdiv(a.Wo); a.div_howo = make_fastdiv(a.Ho * a.Wo); a.div_c = make_fastdiv(a.C); a.div_kw = make_fastdiv(a.kw);
  a.ldx = a.C; a.ldy = a.Co; a.ldm = a.Co; a.relu_co = a.Co;
  return a;
}
}  // namespace

// Label of the kernel instance a forward (which=0) / data-gradient (which=1) call of this descriptor launches
// (matches the demangled name rocprofv3 reports) — used by bench.py to attribute measured time.
extern "C" const char* danhip_conv_kernel_label(const danhip_conv_desc* d, int which) {
  static bf16_t dummy_mask = 0;
  if (!d) return "";
  const bool noscratch = (which & 16) != 0;            // which | 16: the call without a scratch buffer (danhip_conv2d_fwd / _bwd_data): no split-K
  which &= 15;
  const bool masked = which == 5;                      // which = 5: data gradient with the producer's ReLU mask fused (never the library GEMM)
  if (which == 5) which = 1;
  const int cin = (which == 0 || which == 4) ? d->Cin : round_up(d->Cout, 8);
  const int cout = (which == 0 || which == 4) ? d->Cout : d->Cin;
  if (which == 1 && d->stride != 1 && (d->stride & (d->stride - 1)) != 0) return "conv_bwd_data_strided_kernel";
  {
    ConvArgs a = (which == 0 || which == 4) ? fwd_args(d) : bwd_args(d);
    if (which == 4) {                                  // forward conv_relu with the fused 2x2 max-pool (danhip_conv2d_fwd_pool)
      static const float one = 1.f;
      static bf16_t dummy = 0;
      a.bias = &one; a.relu = 1; a.pool_y = &dummy;
      which = 0;
    }
    if (which == 0 && danhip_conv_c8_label(a)) return danhip_conv_c8_label(a);
    const char* cl = danhip_conv_c64_label(a, which == 1);
    if (cl) return cl;
    const bool sk = !noscratch && wants_splitk(a);     // (callers that pass the scratch buffer: dan_amd.ops always does)
    const char* hl = sk ? nullptr : danhip_conv_halo_label(a, which == 1);
    if (hl) return hl;
    if (which == 1) { if (masked) a.mask = &dummy_mask; }
    else { static const float one = 1.f; a.bias = &one; }
    const char* pl = sk ? nullptr : danhip_conv_pointwise_label(a, which == 1);
    if (pl) return pl;
  }
  const bool fast = cin % 64 == 0;
  switch (pick_bn(cout)) {
    case 128: return fast ? "conv_igemm_kernel<128, 128, 2, true>" : "conv_igemm_kernel<128, 128, 2, false>";
    case 64: return fast ? "conv_igemm_kernel<256, 64, 1, true>" : "conv_igemm_kernel<256, 64, 1, false>";
    case 32: return fast ? "conv_igemm_kernel<256, 32, 1, true>" : "conv_igemm_kernel<256, 32, 1, false>";
    default: {
      const long M = which == 0 ? (long)d->N * d->Ho * d->Wo : (long)d->N * d->H * d->W;
      if (M <= 256 * 128) return fast ? "conv_igemm_kernel<64, 16, 1, true>" : "conv_igemm_kernel<64, 16, 1, false>";
      return fast ? "conv_igemm_kernel<256, 16, 1, true>" : "conv_igemm_kernel<256, 16, 1, false>";
    }
  }
}

extern "C" int danhip_pack_conv_weight(const danhip_conv_desc* d, const float* w_hwio, int32_t cin_real, uint16_t* wf_packed,
                                       uint16_t* wb_packed, void* stream) {
  danhip_pack_entry e;
  int32_t blocks = 0;
  int rc = danhip_pack_entry_init(&e, d, w_hwio, cin_real, wf_packed, wb_packed, 0, &blocks);
  if (rc) return rc;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, e);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_pack_entry_init(danhip_pack_entry* e, const danhip_conv_desc* d, const float* w_hwio, int32_t cin_real,
                                      uint16_t* wf_packed, uint16_t* wb_packed, int32_t first_block, int32_t* blocks) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(e && w_hwio && wf_packed && blocks, DANHIP_EINVAL, "pack_entry_init: null pointer");
  DH_REQUIRE(cin_real > 0 && cin_real <= d->Cin, DANHIP_EINVAL, "pack_entry_init: cin_real=%d out of range", cin_real);
  int64_t rf, cf, rb, cb;
  danhip_conv_packed_dims(d, 0, &rf, &cf);
  danhip_conv_packed_dims(d, 1, &rb, &cb);
  e->w_hwio = w_hwio; e->wf_packed = wf_packed; e->wb_packed = wb_packed;
  e->kh = d->kh; e->kw = d->kw; e->cin = d->Cin; e->cin_real = cin_real; e->cout = d->Cout;
  e->rows_f = (int)rf; e->cols_f = (int)cf; e->rows_b = (int)rb; e->cols_b = (int)cb; e->co8 = round_up(d->Cout, 8);
  e->first_block = first_block;
  DH_REQUIRE(rf * cf < (1ll << 31) && rb * cb < (1ll << 31), DANHIP_EINVAL, "pack_entry_init: packed matrix exceeds 2^31 elements");
  const long tiles_f = ((rf + 31) / 32) * (cf / 64);      // forward: 64 k x 32 co tiles (cols_f is a multiple of 64)
  long nbb = wb_packed ? (rb * cb + 1023) / 1024 : 0;      // backward: 4 elements per thread
  if (nbb > 8192) nbb = 8192;
  e->pad_ = (int)tiles_f;
  const long nb = tiles_f + nbb;
  *blocks = (int)nb;
  return DANHIP_OK;
}

extern "C" int danhip_pack_conv_weights_batched(const danhip_pack_entry* table_dev, int32_t n, int32_t total_blocks, void* stream) {
  DH_REQUIRE(table_dev && n > 0 && total_blocks > 0, DANHIP_EINVAL, "pack_conv_weights_batched: bad arguments");
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

// Scratch bytes a forward (which = 0) / data-gradient (which = 1) call of this descriptor can use (split-K partial outputs); 0 = none.
extern "C" size_t danhip_conv2d_workspace_bytes(const danhip_conv_desc* d, int which) {
  if (!d || check_desc(d) != DANHIP_OK) return 0;
  if (which == 1 && d->stride != 1 && (d->stride & (d->stride - 1)) != 0) return 0;
  ConvArgs a = which == 0 ? fwd_args(d) : bwd_args(d);
  int per;
  const int splits = plan_splitk(a, &per);
  if (splits < 2) return 0;
  // shapes the halo / 64->64 / first-layer kernels take never come here with few tiles, except at tiny batch: let them split too
  return (size_t)splits * (size_t)a.M * (size_t)a.Co * sizeof(float);
}

extern "C" int danhip_conv2d_fwd(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, void* y,
                                 int out_dtype, int relu, const uint16_t* residual, void* stream) {
  return danhip_conv2d_fwd_ws(d, x, wf_packed, bias, y, out_dtype, relu, residual, nullptr, 0, stream);
}

extern "C" int danhip_conv2d_fwd_ws(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, void* y,
                                    int out_dtype, int relu, const uint16_t* residual, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(x && wf_packed && y, DANHIP_EINVAL, "conv2d_fwd: null pointer");
  if (out_dtype == DANHIP_F16 && danhip_act_dtype() == DANHIP_F16) out_dtype = DANHIP_BF16;      // alias for "the build's 16-bit type"
  DH_REQUIRE(out_dtype == DANHIP_BF16 || out_dtype == DANHIP_F32 || out_dtype == DANHIP_SPLIT3, DANHIP_EINVAL, "conv2d_fwd: bad out_dtype %d", out_dtype);
  DH_REQUIRE(!(residual && out_dtype != DANHIP_BF16), DANHIP_EINVAL, "conv2d_fwd: residual needs bf16 output");
  if (out_dtype == DANHIP_SPLIT3) {
    DH_REQUIRE(danhip_act_dtype() == DANHIP_F16, DANHIP_EINVAL, "conv2d_fwd: DANHIP_SPLIT3 output needs the fp16 build (IEEE-half limbs)");
    DH_REQUIRE(d->Cout % 8 == 0, DANHIP_EINVAL, "conv2d_fwd: DANHIP_SPLIT3 output needs Cout %% 8 == 0");
    DH_REQUIRE((int64_t)d->N * d->Ho * d->Wo * 3 * d->Cout < (1ll << 31), DANHIP_EINVAL, "conv2d_fwd: the limb-layout output exceeds 2^31 elements");
  }
  ConvArgs a = fwd_args(d);
  a.x = x; a.w = wf_packed; a.bias = bias; a.mask = nullptr; a.resid = residual; a.y = y;
  a.relu = relu; a.out_f32 = (out_dtype == DANHIP_F32); a.accumulate = 0; a.split_out = (out_dtype == DANHIP_SPLIT3);
  if (ws && ws_bytes >= danhip_conv2d_workspace_bytes(d, 0) && ws_bytes > 0) a.splitk_ws = reinterpret_cast<float*>(ws);
  return launch_conv(a, (hipStream_t)stream);
}

// ---- channel-slice views (round 4).  x / y (and the data gradient's mask) may be channel slices of wider NHWC tensors: pointer = base + c0,
// pitch = the wider tensor's channel count.  These calls run on the streaming GEMM / flat-M kernels and, since the second half of round 4,
// on the register-resident 64 -> 64 kernel (the halo kernel addresses dense tensors); 16-bit output, no residual.
static int check_pitch(const danhip_conv_pitch* p, int cx, int cy, const char* what) {
  DH_REQUIRE(p != nullptr, DANHIP_EINVAL, "%s: null pitch", what);
  DH_REQUIRE(p->x_pitch >= cx && p->y_pitch >= cy && (p->aux_pitch == 0 || p->aux_pitch >= cy), DANHIP_EINVAL,
             "%s: a pitch is smaller than the channel count of its tensor", what);
  DH_REQUIRE(((p->x_pitch | p->y_pitch | p->aux_pitch) & 7) == 0, DANHIP_EINVAL, "%s: pitches must be multiples of 8 channels", what);
  return DANHIP_OK;
}

extern "C" int danhip_conv2d_fwd_strided(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                                         int relu, int32_t relu_channels, const danhip_conv_pitch* pitch, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(x && wf_packed && y, DANHIP_EINVAL, "conv2d_fwd_strided: null pointer");
  DH_REQUIRE(d->Cout % 8 == 0, DANHIP_EINVAL, "conv2d_fwd_strided: Cout must be a multiple of 8");
  rc = check_pitch(pitch, d->Cin, d->Cout, "conv2d_fwd_strided");
  if (rc) return rc;
  DH_REQUIRE((int64_t)d->N * d->H * d->W * pitch->x_pitch < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo * pitch->y_pitch < (1ll << 31), DANHIP_EINVAL,
             "conv2d_fwd_strided: a viewed tensor exceeds 2^31 elements");
  DH_REQUIRE(relu_channels >= 0 && relu_channels <= d->Cout && relu_channels % 8 == 0, DANHIP_EINVAL,
             "conv2d_fwd_strided: relu_channels must be a multiple of 8 in [0, Cout]");
  ConvArgs a = fwd_args(d);
  a.x = x; a.w = wf_packed; a.bias = bias; a.y = y;
  a.relu = (relu && relu_channels > 0) ? 1 : 0; a.relu_co = relu_channels;
  a.ldx = pitch->x_pitch; a.ldy = pitch->y_pitch;
  if (ws && ws_bytes >= danhip_conv2d_workspace_bytes(d, 0) && ws_bytes > 0) a.splitk_ws = reinterpret_cast<float*>(ws);
  return launch_conv(a, (hipStream_t)stream);
}

// Forward 1x1 over the channel concatenation of two tensors, never materialised: y = relu([x1 | x2] . W + b), W packed for Cin = c1 + c2.
// Only the streaming GEMM (conv_pointwise.hip) walks two sources: _supported() says whether this shape takes it (64-multiple channel
// counts on both sides, at least 2048 output pixels); the host concatenates and calls danhip_conv2d_fwd otherwise.
static int concat2_args(const danhip_conv_desc* d, int32_t c1, int32_t src_pitch, ConvArgs& a) {
  if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->H != d->Ho || d->W != d->Wo) return 0;
  if (c1 <= 0 || c1 >= d->Cin || c1 % 64 != 0 || (d->Cin - c1) % 64 != 0 || d->Cout % 64 != 0) return 0;
  if (src_pitch % 8 != 0 || src_pitch < c1 || src_pitch < d->Cin - c1) return 0;
  if ((int64_t)d->N * d->H * d->W * src_pitch >= (1ll << 31)) return 0;
  a = fwd_args(d);
  a.ldx = src_pitch;
  a.ksplit = c1 / 64;
  return 1;
}

extern "C" int danhip_conv2d_fwd_concat2_supported(const danhip_conv_desc* d, int32_t c1, int32_t src_pitch) {
  if (!d || check_desc(d) != DANHIP_OK) return 0;
  ConvArgs a{};
  if (!concat2_args(d, c1, src_pitch, a)) return 0;
  static const float one = 1.f;
  static bf16_t dummy = 0;
  a.bias = &one; a.x2 = &dummy;
  return danhip_conv_pointwise_label(a, false) != nullptr ? 1 : 0;
}

extern "C" int danhip_conv2d_fwd_concat2(const danhip_conv_desc* d, const uint16_t* x1, const uint16_t* x2, int32_t c1, int32_t src_pitch,
                                         const uint16_t* wf_packed, const float* bias, uint16_t* y, int relu, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(x1 && x2 && wf_packed && y, DANHIP_EINVAL, "conv2d_fwd_concat2: null pointer");
  ConvArgs a{};
  DH_REQUIRE(concat2_args(d, c1, src_pitch, a), DANHIP_EINVAL,
             "conv2d_fwd_concat2: 1x1 / stride 1 with 64-multiple channel counts (c1, Cin - c1, Cout) and an 8-multiple source pitch only");
  a.x = x1; a.x2 = x2; a.w = wf_packed; a.bias = bias; a.y = y; a.relu = relu ? 1 : 0;
  rc = danhip_launch_conv_pointwise(a, (hipStream_t)stream);
  DH_REQUIRE(rc <= 0, DANHIP_EINVAL, "conv2d_fwd_concat2: shape not taken by the streaming GEMM (ask danhip_conv2d_fwd_concat2_supported first)");
  return rc;
}

extern "C" int danhip_conv2d_bwd_data_strided(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint16_t* relu_mask,
                                              uint16_t* dx, int accumulate, const danhip_conv_pitch* pitch, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(dy && wb_packed && dx, DANHIP_EINVAL, "conv2d_bwd_data_strided: null pointer");
  DH_REQUIRE(d->stride == 1, DANHIP_EINVAL, "conv2d_bwd_data_strided: stride 1 only");
  const int co8 = round_up(d->Cout, 8);
  rc = check_pitch(pitch, co8, d->Cin, "conv2d_bwd_data_strided");
  if (rc) return rc;
  DH_REQUIRE((int64_t)d->N * d->Ho * d->Wo * pitch->x_pitch < (1ll << 31) && (int64_t)d->N * d->H * d->W * pitch->y_pitch < (1ll << 31), DANHIP_EINVAL,
             "conv2d_bwd_data_strided: a viewed tensor exceeds 2^31 elements");
  ConvArgs a = bwd_args(d);
  a.x = dy; a.w = wb_packed; a.mask = relu_mask; a.y = dx; a.accumulate = accumulate;
  a.ldx = pitch->x_pitch; a.ldy = pitch->y_pitch; a.ldm = pitch->aux_pitch ? pitch->aux_pitch : d->Cin;
  if (ws && ws_bytes >= danhip_conv2d_workspace_bytes(d, 1) && ws_bytes > 0) a.splitk_ws = reinterpret_cast<float*>(ws);
  return launch_conv(a, (hipStream_t)stream);
}

// conv_relu followed by tf.layers.max_pooling2d([2,2],[2,2],'same') (net/sfd_net.py:128-143: every VGG block): y as
// danhip_conv2d_fwd(relu = 1) and pool_y = maxpool2x2(y) [N,ceil(Ho/2),ceil(Wo/2),Cout].  The 3x3 kernels that own whole row
// pairs per wave pool their packed outputs in the epilogue (no second pass over y); other shapes run the pool kernel after.
bool danhip_conv_pool_fusable(const ConvArgs& a) {
  if (!(a.bias && a.relu && !a.resid && !a.out_f32 && !a.mask && !a.accumulate)) return false;
  return danhip_conv_c64_eligible(a) || danhip_conv_halo_pool_fusable(a);
}

extern "C" int danhip_conv2d_fwd_pool(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                                      uint16_t* pool_y, void* stream) {
  return danhip_conv2d_fwd_pool_arg(d, x, wf_packed, bias, y, pool_y, nullptr, stream);
}

extern "C" int danhip_conv2d_fwd_pool_arg(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                                          uint16_t* pool_y, uint8_t* pool_arg, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(x && wf_packed && bias && pool_y, DANHIP_EINVAL, "conv2d_fwd_pool: null pointer");
  DH_REQUIRE(d->Cout % 8 == 0, DANHIP_EINVAL, "conv2d_fwd_pool: Cout must be a multiple of 8");
  ConvArgs a = fwd_args(d);
  a.x = x; a.w = wf_packed; a.bias = bias; a.mask = nullptr; a.resid = nullptr; a.y = y;
  a.relu = 1; a.out_f32 = 0; a.accumulate = 0;
  const bool fused = danhip_conv_pool_fusable(a);
  // y == NULL: only the pooled map is wanted (inference: nothing reads the full-resolution activation of conv1_2 / conv2_2) - the fusing
  // kernels then skip its stores; ask danhip_conv2d_fwd_pool_only(d) first
  DH_REQUIRE(y || fused, DANHIP_EINVAL, "conv2d_fwd_pool: y == NULL needs a kernel that pools in its epilogue (danhip_conv2d_fwd_pool_only)");
  a.pool_y = fused ? pool_y : nullptr;
  a.pool_arg_out = fused ? pool_arg : nullptr;
  rc = launch_conv(a, (hipStream_t)stream);
  if (rc || fused) return rc;
  return danhip_maxpool2x2_fwd_arg(y, pool_y, pool_arg, d->N, d->Ho, d->Wo, d->Cout, stream);
}

extern "C" int danhip_conv2d_fwd_pool_only(const danhip_conv_desc* d) {
  if (!d || check_desc(d) != DANHIP_OK || d->Cout % 8 != 0) return 0;
  static const float one = 1.f;
  static bf16_t dummy = 0;
  ConvArgs a = fwd_args(d);
  a.bias = &one; a.relu = 1; a.pool_y = &dummy;
  if (!danhip_conv_pool_fusable(a)) return 0;
  if (danhip_conv_c64_eligible(a)) return 1;
  return (int64_t)a.N * a.H * a.W * a.Co * 2 <= (1ll << 31) ? 1 : 0;      // (the halo kernel's lean epilogue: the one with descriptor stores)
}

// conv_relu (+ the fused 2x2 max-pool when pool_y is given) that ALSO writes the ReLU bit masks of its outputs (danhip_relu_bits layout) from
// the packed values still in registers — for the data gradient of the NEXT convolution (danhip_conv2d_bwd_data_bits).  Only the
// 128-wide halo tiles do this: ask danhip_conv2d_fwd_emits_bits first.
extern "C" int danhip_conv2d_fwd_emits_bits(const danhip_conv_desc* d, int with_pool) {
  if (!d || check_desc(d) != DANHIP_OK || d->Cout % 8 != 0) return 0;
  static const float one = 1.f;
  static bf16_t dummy = 0;
  ConvArgs a = fwd_args(d);
  a.bias = &one; a.relu = 1;
  if (danhip_conv_c8_label(a)) return with_pool ? 0 : 1;        // the first layer's store-bound kernel writes the mask beside its output
  if (danhip_conv_c64_eligible(a)) return 0;
  if (!danhip_conv_halo_emits_bits(a)) return 0;
  if (with_pool) { a.pool_y = &dummy; if (!danhip_conv_halo_pool_fusable(a)) return 0; }
  return 1;
}

extern "C" int danhip_conv2d_fwd_relu_bits(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                                           uint8_t* y_bits, uint16_t* pool_y, uint8_t* pool_bits, void* stream) {
  return danhip_conv2d_fwd_relu_bits_arg(d, x, wf_packed, bias, y, y_bits, pool_y, pool_bits, nullptr, stream);
}

extern "C" int danhip_conv2d_fwd_relu_bits_arg(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                                               uint8_t* y_bits, uint16_t* pool_y, uint8_t* pool_bits, uint8_t* pool_arg, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  // y == NULL (with pool_y): only the pooled map, the bit masks and the arg-max codes are wanted - the halo kernel's lean epilogue drops the
  // full-resolution stores (training: nothing but the pool reads conv2_2's output; ask danhip_conv2d_fwd_pool_only(d) first)
  DH_REQUIRE(x && wf_packed && bias && y_bits && (!pool_y == !pool_bits), DANHIP_EINVAL, "conv2d_fwd_relu_bits: null pointer");
  DH_REQUIRE(y || (pool_y && danhip_conv2d_fwd_pool_only(d)), DANHIP_EINVAL,
             "conv2d_fwd_relu_bits: y == NULL needs a kernel that pools in its epilogue (danhip_conv2d_fwd_pool_only)");
  DH_REQUIRE(danhip_conv2d_fwd_emits_bits(d, pool_y != nullptr), DANHIP_EINVAL,
             "conv2d_fwd_relu_bits: this shape's forward kernel does not write bit masks (ask danhip_conv2d_fwd_emits_bits)");
  ConvArgs a = fwd_args(d);
  a.x = x; a.w = wf_packed; a.bias = bias; a.mask = nullptr; a.resid = nullptr; a.y = y;
  a.relu = 1; a.out_f32 = 0; a.accumulate = 0;
  a.pool_y = pool_y; a.bits_out = y_bits; a.pool_bits_out = pool_bits; a.pool_arg_out = pool_y ? pool_arg : nullptr;
  if (danhip_conv_c8_label(a)) return danhip_launch_conv_c8(a, (hipStream_t)stream);
  return danhip_launch_conv_halo(a, (hipStream_t)stream);
}

namespace {
// Direct (gather-form) data gradient for strided convolutions (only conv6_2 / conv7_2, 3x3 stride 2, tiny maps).
// One thread per (input pixel, 8 input channels).
__global__ void conv_bwd_data_strided_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wb, const bf16_t* __restrict__ mask,
                                             bf16_t* __restrict__ dx, int N, int H, int W, int Cin, int Ho, int Wo, int co8, int kh, int kw,
                                             int stride, int pad_t, int pad_l, int kpad_b, int accumulate) {
  const long total = (long)N * H * W * (Cin / 8);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % (Cin / 8));
    long pix = idx / (Cin / 8);
    const int wi = (int)(pix % W); pix /= W;
    const int hi = (int)(pix % H);
    const int n = (int)(pix / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < kh; ++i) {
      const int hn = hi + pad_t - i;
      if (hn < 0 || hn % stride) continue;
      const int ho = hn / stride;
      if (ho >= Ho) continue;
      for (int j = 0; j < kw; ++j) {
        const int wn = wi + pad_l - j;
        if (wn < 0 || wn % stride) continue;
        const int wo = wn / stride;
        if (wo >= Wo) continue;
        const bf16_t* g = dy + ((long)(n * Ho + ho) * Wo + wo) * co8;
        const int tapf = (kh - 1 - i) * kw + (kw - 1 - j);      // wb stores flipped taps
        for (int co = 0; co < co8; ++co) {
          const float gv = bf2f(g[co]);
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[r] += gv * bf2f(wb[(long)(cg * 8 + r) * kpad_b + tapf * co8 + co]);
        }
      }
    }
    const long o = ((long)(n * H + hi) * W + wi) * Cin + cg * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = acc[r];
      if (mask && !(bf2f(mask[o + r]) > 0.f)) v = 0.f;
      if (accumulate) v += bf2f(dx[o + r]);
      dx[o + r] = f2bf(v);
    }
  }
}
}  // namespace

extern "C" int danhip_conv2d_bwd_data(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint16_t* relu_mask,
                                      uint16_t* dx, int accumulate, void* stream) {
  return danhip_conv2d_bwd_data_ws(d, dy, wb_packed, relu_mask, dx, accumulate, nullptr, 0, stream);
}

extern "C" int danhip_conv2d_bwd_data_ws(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint16_t* relu_mask,
                                         uint16_t* dx, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(dy && wb_packed && dx, DANHIP_EINVAL, "conv2d_bwd_data: null pointer");
  const int co8 = round_up(d->Cout, 8);
  const int taps = d->kh * d->kw;
  const int pad_t = same_pad_before(d->H, d->Ho, d->kh, d->stride), pad_l = same_pad_before(d->W, d->Wo, d->kw, d->stride);
  const bool pow2 = (d->stride & (d->stride - 1)) == 0;
  if (d->stride != 1 && !pow2) {
    const long total = (long)d->N * d->H * d->W * (d->Cin / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(conv_bwd_data_strided_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, wb_packed, relu_mask, dx, d->N,
                       d->H, d->W, d->Cin, d->Ho, d->Wo, co8, d->kh, d->kw, d->stride, pad_t, pad_l, round_up(taps * co8, 64), accumulate);
    DH_LAUNCH_CHECK();
    return DANHIP_OK;
  }
  ConvArgs a = bwd_args(d);
  a.x = dy; a.w = wb_packed; a.bias = nullptr; a.mask = relu_mask; a.resid = nullptr; a.y = dx;
  a.relu = 0; a.out_f32 = 0; a.accumulate = accumulate;
  if (ws && ws_bytes >= danhip_conv2d_workspace_bytes(d, 1) && ws_bytes > 0) a.splitk_ws = reinterpret_cast<float*>(ws);
  return launch_conv(a, (hipStream_t)stream);
}

extern "C" int danhip_conv2d_bwd_data_takes_bits(const danhip_conv_desc* d) {
  if (!d || check_desc(d) != DANHIP_OK) return 0;
  if (d->stride != 1) return 0;
  ConvArgs a = bwd_args(d);
  a.bias = nullptr; a.relu = 0; a.out_f32 = 0; a.resid = nullptr;
  if (danhip_conv_c64_eligible(a)) return (a.W % 2 == 0) ? 1 : 0;   // 64 -> 64 register-resident kernel: 2 KiB of bits per tile through LDS (pixel pairs)
  return danhip_conv_halo_takes_bits(a) ? 1 : 0;
}

// conv1_2's data gradient with conv1_1's weight / bias gradient folded in (conv_halo_c64.hip FUSE8): dX is never written.
extern "C" int danhip_conv2d_bwd_data_first_supported(const danhip_conv_desc* d) {
  if (!d || check_desc(d) != DANHIP_OK || d->stride != 1) return 0;
  ConvArgs a = bwd_args(d);
  static const unsigned char dummy = 0;
  a.mask_bits = &dummy;
  return (danhip_conv_c64_eligible(a) && !a.strided() && a.W % 2 == 0 && (int64_t)a.N * a.H * a.W * 16 < (1ll << 31)) ? 1 : 0;
}

extern "C" int danhip_conv2d_bwd_data_bits_first(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint8_t* relu_bits,
                                                 const uint16_t* x8, int32_t cin_real, float* dw8, float* db8, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(dy && wb_packed && relu_bits && x8 && dw8, DANHIP_EINVAL, "conv2d_bwd_data_bits_first: null pointer");
  DH_REQUIRE(cin_real >= 1 && cin_real <= 4, DANHIP_EINVAL, "conv2d_bwd_data_bits_first: cin_real = %d (1..4 real channels of the 8-channel image)", cin_real);
  DH_REQUIRE(danhip_conv2d_bwd_data_first_supported(d), DANHIP_EINVAL, "conv2d_bwd_data_bits_first: not the 64 -> 64 3x3 shape the folded kernel takes "
             "(ask danhip_conv2d_bwd_data_first_supported first)");
  ConvArgs a = bwd_args(d);
  a.x = dy; a.w = wb_packed; a.bias = nullptr; a.mask = nullptr; a.mask_bits = relu_bits; a.resid = nullptr; a.y = nullptr;
  a.relu = 0; a.out_f32 = 0; a.accumulate = 0;
  a.fuse_x8 = x8; a.fuse_dw = dw8; a.fuse_db = db8; a.fuse_cin_real = cin_real;
  rc = danhip_launch_conv_c64(a, (hipStream_t)stream);
  DH_REQUIRE(rc <= 0, DANHIP_EINVAL, "conv2d_bwd_data_bits_first: the folded kernel declined the call");
  return rc;
}

extern "C" int danhip_conv2d_bwd_data_bits(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint8_t* relu_bits,
                                           uint16_t* dx, int accumulate, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  DH_REQUIRE(dy && wb_packed && dx && relu_bits, DANHIP_EINVAL, "conv2d_bwd_data_bits: null pointer");
  DH_REQUIRE(danhip_conv2d_bwd_data_takes_bits(d), DANHIP_EINVAL, "conv2d_bwd_data_bits: the data-gradient kernel of this shape takes the 16-bit mask "
             "(ask danhip_conv2d_bwd_data_takes_bits first)");
  ConvArgs a = bwd_args(d);
  a.x = dy; a.w = wb_packed; a.bias = nullptr; a.mask = nullptr; a.mask_bits = relu_bits; a.resid = nullptr; a.y = dx;
  a.relu = 0; a.out_f32 = 0; a.accumulate = accumulate;
  if (danhip_conv_c64_eligible(a)) return danhip_launch_conv_c64(a, (hipStream_t)stream);
  return danhip_launch_conv_halo(a, (hipStream_t)stream);
}
